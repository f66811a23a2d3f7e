#!/usr/bin/env python3
"""Where does a harness LM iteration spend its time on the GPU box? (diagnostic)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from voxgraph_amd import capi
capi.load()
import torch
from harness import lm
from harness.backends import GpuBackend
from threadpoolctl import threadpool_limits
from scipy.linalg import solveh_banded

class A: pass
args = A(); args.grid=[20,10]; args.block_dims=[16,16,16]; args.block_min=[-8,-8,-4]; args.voxel_size=0.2
args.truncation=0.6; args.esdf_max=2.0; args.pose_sigma=0.3; args.yaw_sigma=0.05; args.seed=2
true_poses, poses, pairs = bench.build_graph(args)
ctx = capi.Context(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
subs=[]
for k in range(len(true_poses)):
    sm = capi.Submap.synth_city(ctx, k, 0.2, 16, args.block_min, args.block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
    sm.extract_voxel_points(1.0, 0.3, True); sm.release_raw_layers(); subs.append(sm)
cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
cfs=[capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a,b in pairs]
batch = capi.RegistrationBatch(ctx, cfs, pairs)
n=len(true_poses)
backend = GpuBackend(capi, ctx, batch, n)
def T(f, k=10):
    f(); t=time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter()-t)/k*1e3
print("backend() ms", T(lambda: backend(poses)))
print(" evaluate_normal only ms", T(lambda: (batch.evaluate_normal(poses, to_host=False), ctx.synchronize())))
print(" assemble only ms", T(lambda: (batch.assemble(n, backend.buf.data_ptr()), ctx.synchronize())))
print(" host copy ms", T(lambda: (backend.host.copy_(backend.buf, non_blocking=True), torch.cuda.current_stream().synchronize())))
buf = np.array(backend(poses))
info=[1.0,1.0,2500.0,2500.0]
edges=[lm.RelativePoseEdge.from_poses(k,k+1,poses[k],poses[k+1],info) for k in range(n-1)]
prob = lm.Problem(lambda p: buf, n, pairs, edges)
t=time.perf_counter(); prob._prepare_reduced(); print("prepare_reduced ms", (time.perf_counter()-t)*1e3)
print("evaluate_reduced (no backend) ms", T(lambda: prob.evaluate_reduced(poses)))
c,g,(band,V)=prob.evaluate_reduced(poses)
Ab=band.copy(); Ab[prob._u]+=np.clip(band[prob._u],1e-6,1e32)/1e4
print("banded solve, default threads ms", T(lambda: solveh_banded(Ab,g,lower=False,check_finite=False)))
t=time.perf_counter()
with threadpool_limits(4):
    print("enter limits ms", (time.perf_counter()-t)*1e3)
    print("banded solve, 4 threads ms", T(lambda: solveh_banded(Ab,g,lower=False,check_finite=False)))
    print("evaluate_reduced 4 threads ms", T(lambda: prob.evaluate_reduced(poses)))
    print("matvec ms", T(lambda: prob.reduced_matvec(V,g)))
    t=time.perf_counter(); x,s = lm._solve(lm.Problem(backend,n,pairs,edges), poses, 1e-10,1e-6,1e-10,50,1e9,1e4,False)
    print("full _solve ms", (time.perf_counter()-t)*1e3, s["evaluations"])
