"""MIOpen solver selection for the encoders' convolutions.

This image ships no gfx950 find-db, so out of the box MIOpen's immediate mode runs its naive fall-back kernels on the
first call of every configuration (first step ~80 s) and then one heuristic choice per configuration.  The files under
`pixpro-with-opticalflow_amd/miopen_db/` are MIOpen's own user find-db / perf-db / kernel cache, written by MIOpen
itself during one `torch.backends.cudnn.benchmark = True` run of `bench.py` on an MI355X (462 s of tuning for the
ResNet-50, 224^2, batch-256 configurations; tools/README in DESIGN.md 5).  Pointing MIOpen at them makes it pick the
measured-fastest solver per configuration at once: 246 -> 228 ms per step.  Configurations that are not in the files
behave exactly as before.  A user's own MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR is never overridden.
"""
import os

DB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")


def use_shipped_db():
    """Call before the first convolution.  Returns True when the shipped files are in use."""
    if os.environ.get("PIXPRO_NO_MIOPEN_DB") == "1" or not os.path.isdir(DB_DIR):
        return False
    if not any(f.endswith(".ufdb.txt") for f in os.listdir(DB_DIR)) or not os.access(DB_DIR, os.W_OK):
        return False
    os.environ.setdefault("MIOPEN_USER_DB_PATH", DB_DIR)
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", DB_DIR)
    return os.environ["MIOPEN_USER_DB_PATH"] == DB_DIR
