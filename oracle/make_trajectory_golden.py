"""TEST INFRASTRUCTURE: writes tests/golden/trajectories.npz -- the fp32 CPU-oracle DDIM trajectories that
tests/test_hip_trajectory.py compares the GPU path with at the BASELINE shapes:

  c2  512x512, 50 CFG steps, zero unconditional context (BASELINE configs[1]); latent, first step, decoded image
  c5  768x768, the 31 real steps of the "30-step" schedule, non-zero unconditional context (configs[4]); latent, first step
  c3  ControlNet + SeeCoder-PA + control hint, 512x512, 10 steps (configs[2]); latent, first step with / without control

They are outputs of oracle/pfd_oracle.py (tests/oracle_worker.py holds the three cases) on the seeded weights of
oracle/weights.py -- 13 to 18 minutes of host time (8 to 64 threads), which is why they are fixtures: run in line they made the GPU
suite 21 minutes long, and run next to it they starve the suite's other CPU-oracle checks (the GPU boxes give a container
about 64 threads' worth of CPU).  The oracle itself is pinned to the reference by tests/golden/golden.npz
(oracle/make_golden.py imports the reference); THESE fixtures are pinned to the oracle by
tests/test_oracle_golden.py::test_trajectory_fixture_first_and_last_step, which recomputes the first DDIM step of every case
and the LAST step from the stored penultimate latent on the CPU in the `-m "not gpu"` suite and asserts the digest of the
oracle sources stored in the meta (a changed oracle means: regenerate), and `PFD_ORACLE_LIVE=1 pytest -m gpu` recomputes the whole trajectories instead of
reading them.

    python oracle/make_trajectory_golden.py [--threads N] [--from-dir DIR]     (DIR: <case>.pt files oracle_worker.py wrote)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CASES = ("c2", "c5", "c3")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--from-dir", default=None)
    args = ap.parse_args()
    d = args.from_dir or tempfile.mkdtemp(prefix="pfd_traj_")
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_worker as OW
    out, meta = {}, {"script": "oracle/make_trajectory_golden.py", "torch": torch.__version__,
                     "host_cpus": os.cpu_count(), "written": time.strftime("%Y-%m-%d"), "cases": {},
                     "oracle_sources": list(OW.ORACLE_SOURCES), "oracle_sha256": OW.oracle_digest()}
    for c in CASES:
        pt = os.path.join(d, c + ".pt")
        if not os.path.exists(pt):
            subprocess.run([sys.executable, os.path.join(REPO, "tests", "oracle_worker.py"), c, pt, str(args.threads)], check=True)
        r = torch.load(pt)
        meta["cases"][c] = {"steps": int(r["steps"]), "seconds": round(float(r["seconds"]), 1), "threads": int(r["threads"])}
        for k, v in r.items():
            if torch.is_tensor(v):
                # the decoded image is compared at 2e-2: fp16 (5e-4) keeps the file small; latents stay fp32
                out[f"{c}.{k}"] = v.numpy().astype(np.float16 if k == "image" else np.float32)
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(REPO, "tests", "golden", "trajectories.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB", json.dumps(meta))


if __name__ == "__main__":
    main()
