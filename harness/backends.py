"""Registration backends for harness/lm.py: the GPU library (product) and the CPU
oracle (checker), both returning the fused normal-equation buffer."""
import numpy as np


class GpuBackend:
    """One solver evaluation of the registration constraints of this rank's shard:
    vgx_reg_batch_evaluate_normal -> vgx_reg_batch_scatter_normal -> ONE all-reduce(sum, the words taken as
    int64) of the [n_global][45] array of per-constraint blocks (RCCL under torch.distributed) -> vgx_reg_assembler_assemble.
    Every row of the array is written by exactly one rank, so the sum is exact in any order and every rank
    assembles the single-GPU buffer bit for bit, whatever the number of ranks (include/voxgraph_amd.h).
    Without torch.distributed the batch holds the whole list and assembles directly."""

    def __init__(self, capi, ctx, batch, n_nodes, dist=None, node_pair_global=None):
        import torch
        self.torch, self.ctx, self.batch, self.n_nodes, self.dist = torch, ctx, batch, n_nodes, dist
        self.buf = torch.zeros(capi.fused_size(n_nodes, batch.n_global), dtype=torch.float64,
                               device="cuda")
        self.host = torch.zeros_like(self.buf, device="cpu").pin_memory()
        self.sharded = dist is not None and dist.is_initialized()
        if self.sharded:
            if node_pair_global is None or len(node_pair_global) != batch.n_global:
                raise ValueError("a sharded backend needs the whole list's node pairs (node_pair_global)")
            self.assembler = capi.RegistrationAssembler(ctx, node_pair_global)
            self.blocks = torch.zeros((batch.n_global, capi.NORMAL_SIZE), dtype=torch.float64, device="cuda")
        # the zero-fill above ran on torch's stream; the library writes the buffer on ITS stream
        torch.cuda.current_stream().synchronize()

    def _library_stream_is_torchs(self):
        return self.ctx.get_stream() == self.torch.cuda.current_stream().cuda_stream

    def __call__(self, poses):
        self.batch.evaluate_normal(poses, to_host=False)
        if self.sharded:
            self.batch.scatter_normal(self.blocks.data_ptr(), zero_first=True)
            # the all-reduce is ordered after torch's current stream; the scatter ran on the
            # library's stream.  Same stream (ctx.set_stream(torch's), as bench.py does): stream
            # order suffices.  Otherwise the array must be complete before RCCL reads it.
            same = self._library_stream_is_torchs()
            if not same:
                self.ctx.synchronize()
            # summed as int64 words: a word is non-zero on exactly one rank, so the sum is that rank's BIT PATTERN
            # whatever order the collective adds in (an f64 sum would also turn a -0.0 into +0.0)
            self.dist.all_reduce(self.blocks.view(self.torch.int64))
            if not same:
                self.torch.cuda.current_stream().synchronize()
            self.assembler.assemble(self.blocks.data_ptr(), self.n_nodes, self.buf.data_ptr())
        else:
            self.batch.assemble(self.n_nodes, self.buf.data_ptr(), zero_first=True)
        # order the copy after the library's kernels (no-op wait when everything is drained)
        self.ctx.synchronize()
        self.host.copy_(self.buf, non_blocking=True)
        self.torch.cuda.current_stream().synchronize()
        return self.host.numpy()


def assemble_fused(normals, pairs, n_nodes, n_global=None, global_index=None):
    """numpy restatement of vgx_reg_batch_assemble's layout (test infrastructure)."""
    n_global = len(pairs) if n_global is None else n_global
    buf = np.zeros(1 + 20 * n_nodes + 16 * n_global)
    iu = np.triu_indices(8)
    for c, (a, b) in enumerate(pairs):
        nb = normals[c]
        H = np.zeros((8, 8))
        H[iu] = nb[9:]
        H = H + np.triu(H, 1).T
        buf[0] += nb[0]
        for side, node in enumerate((a, b)):
            buf[1 + 4 * node:5 + 4 * node] += nb[1 + 4 * side:5 + 4 * side]
            o = 1 + 4 * n_nodes + 16 * node
            buf[o:o + 16] += H[4 * side:4 * side + 4, 4 * side:4 * side + 4].ravel()
        g = c if global_index is None else global_index[c]
        o = 1 + 20 * n_nodes + 16 * g
        buf[o:o + 16] = H[0:4, 4:8].ravel()
    return buf


class OracleBackend:
    """CPU oracle per constraint (oracle/reg_oracle.c), one constraint per task."""

    def __init__(self, layers, points, pairs, n_nodes, threads=4):
        from concurrent.futures import ThreadPoolExecutor
        self.layers, self.points, self.pairs, self.n_nodes = layers, points, list(pairs), n_nodes
        self.pool = ThreadPoolExecutor(threads)      # pose_graph.cpp:96 num_threads = 4

    def __call__(self, poses):
        from oracle import pyoracle as orc

        def one(c):
            a, b = self.pairs[c]
            xyz, dist, w = self.points[a]
            ok, cost, jtr, jtj = orc.reg_evaluate_normal(self.layers[b], xyz, dist, w, poses[a], poses[b])
            assert ok
            return np.concatenate([[cost], jtr, jtj])

        normals = list(self.pool.map(one, range(len(self.pairs))))
        return assemble_fused(normals, self.pairs, self.n_nodes)
