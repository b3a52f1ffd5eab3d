import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, starky_bn254_amd as S, oracle_lib as O
for name, mk, inp in (("g1", lambda: S.G1ExpStark(128), lambda: O.g1exp_inputs(128, 1)[0]), ("g2", lambda: S.G2ExpStark(128), lambda: O.g2exp_inputs(128, 2)[0])):
    stark = mk(); cfg = stark.config(); p = S.Prover(stark, cfg, 16); ios = inp()
    p.generate_trace(ios); ts = []
    for _ in range(5):
        p.generate_trace(ios); ts.append(p.stage_times()["device_tracegen_ms"])
    print(name, "device_tracegen_ms", [round(t, 2) for t in ts], flush=True)
    p.close()
