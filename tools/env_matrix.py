"""Plan-time knobs against one another (round 5): U11L-128 under 13 combinations of the diagnostic environment variables, batches of 1 ... 700
rows: error against the float64 oracle, and whether the bits equal the default plan's (they must, except where a knob changes the arithmetic's
shape: HIGSFA_NO_REM4 / HIGSFA_NO_PACK).  python tools/env_matrix.py   (profiles/r05_env_matrix.txt)"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/pyfaceanalysis_amd") else os.getcwd())
sys.path.insert(0, os.getcwd())
from pyfaceanalysis_amd import synth
from pyfaceanalysis_amd.flow import Flow
from oracle import mdp_restate as oracle
blob, nodes = synth.cached_preset_blob("U11L-128")
x8 = synth.make_subimages(700, 128, dtype=np.uint8)
ref = oracle.execute_flow(nodes, x8[:48].astype(np.float64))
base = Flow.from_blob(blob, device=0, output_dtype=np.float32)
b = base.execute(x8)
envs = [{"HIGSFA_NO_REM4": "1"}, {"HIGSFA_NO_PACK": "1"}, {"HIGSFA_NO_DIRECT": "1"}, {"HIGSFA_NO_FSPEC": "1", "HIGSFA_NO_DIRECT": "1"}, {"HIGSFA_TAIL": "0"}, {"HIGSFA_TAIL": "1"},
        {"HIGSFA_TAIL": "2"}, {"HIGSFA_NO_SOA": "1"}, {"HIGSFA_SUBTREE_WGS": "100000", "HIGSFA_SUBTREE": "100000"}, {"HIGSFA_NO_REM4": "1", "HIGSFA_NO_PACK": "1", "HIGSFA_SUBTREE_WGS": "100000"},
        {"HIGSFA_TAIL": "1", "HIGSFA_SUBTREE_WGS": "100000"}, {"HIGSFA_SPLITM_WGS": "100000"}, {"HIGSFA_SPLITM_MAX": "0", "HIGSFA_SPLITM_WGS": "0"}]
for env in envs:
    for k, v in env.items(): os.environ[k] = v
    f = Flow.from_blob(blob, device=0, output_dtype=np.float32)
    d = f.describe()
    res = []
    for n in (1, 18, 130, 348, 600, 700):
        y = f.execute(x8[:n])
        err = float(np.abs(y[:min(n, 48)].astype(np.float64) - ref[:min(n, 48)]).max() / np.abs(ref).max())
        res.append((n, bool(np.array_equal(y, b[:n])), "%.1e" % err))
    for k in env: del os.environ[k]
    runs = [ln.split("[batches of up to")[1][:50] for ln in d.splitlines() if "sub-trees in ONE launch" in ln]
    worst = max(float(r[2]) for r in res)
    print(env, "same bits as default:", [r[1] for r in res], "worst err %.1e" % worst, "OK" if worst <= 1e-4 else "FAIL", runs)
    f.close()
