"""Rank-0 console logging used by the model constructors (reference lib/log_service.py:13-35).
Unlike the reference it does not divide by torch.cuda.device_count() (lib/sync.py:31-35), so it
also works on a GPU-less host; the rank comes from torch.distributed / the launcher env."""
import os


def _local_rank():
    for k in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"):
        if k in os.environ:
            try:
                return int(os.environ[k])
            except ValueError:
                pass
    return 0


def print_log(*console_info):
    if _local_rank() != 0:
        return
    line = " ".join(str(i) for i in console_info)
    if os.environ.get("PFD_QUIET", "0") != "1":
        print(line)
    log_file = None
    try:
        from .cfg_holder import cfg_unique_holder as cfguh
        cfg = cfguh().cfg
        log_file = cfg.train.log_file if "train" in cfg else cfg.eval.log_file
    except Exception:
        return
    if log_file is not None:
        with open(log_file, "a") as f:
            f.write(line + "\n")
