"""Builds the oracle's C/OpenMP twin (test infrastructure) into oracle/_build/libvi_oracle.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "vi_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libvi_oracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-o", OUT, SRC, "-lm"],
                   check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
