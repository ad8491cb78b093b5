#!/usr/bin/env python3
"""CLI of mujoco_sim_amd/_isa_lint.py (the check every build of libmjsim.so runs on its device assembly): isa_lint.py file.s [...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mujoco_sim_amd._isa_lint import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
