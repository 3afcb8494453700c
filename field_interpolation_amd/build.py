"""Builds libfi_hip.so (gfx950) in-tree with hipcc.  `python -m field_interpolation_amd.build`."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfi_hip.so")


def build(force=False, jobs=4):
    csrc = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-C", csrc, "-j%d" % jobs])
    if not os.path.exists(LIB):
        raise RuntimeError("libfi_hip.so was not produced")
    return LIB


if __name__ == "__main__":
    print(build())
