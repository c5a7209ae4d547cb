"""Which files under tests/golden hold a stage dump (problem tables + manifest + reference stage outputs): the trajectory, frame and
full-size ("slim") fixtures carry step logs, states or output frames only and belong to the scene tests."""
import glob
import os
import zipfile

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def stage_dumps():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        with zipfile.ZipFile(p) as z:
            names = set(z.namelist())
        if "manifest_json.npy" in names and "geometry" not in os.path.basename(p) and not os.path.basename(p).startswith("slim_"):
            out.append(p)
    return out
