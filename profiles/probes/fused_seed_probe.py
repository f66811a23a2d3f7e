import os, sys
sys.path.insert(0, "/root/repo")
os.environ["SEEDS"]="1"; os.environ["FIRST"]="300290"
import numpy as np
import profiles.fuzz_reg as fz
# monkeypatch: capture inside by re-running the body with prints: simplest is to copy the check
src = open("/root/repo/profiles/fuzz_reg.py").read()
src = src.replace("                mag = np.r_[", "                print('normal', normal[0][1:9]); print('want  ', want[1:9]); print('diff  ', normal[0][1:9]-want[1:9]); print('mag   ', (np.abs(J).T @ np.abs(r0))); print('n with J', int((np.abs(jo0).sum(1)>0).sum()), 'of', n, 'poses', poses)\n                mag = np.r_[")
exec(compile(src, "fuzz_reg_dbg", "exec"))
