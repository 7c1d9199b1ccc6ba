#!/usr/bin/env python3
"""Drop-in for the reference's `python infer.py PATH -ckpt ... -c ...` (same options, same .lab output), running the
MI355X path.  See wfl-asr_amd/infer.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from wfl_asr_amd.infer import infer_audio, infer_folder, load_config, main  # noqa: E402,F401

if __name__ == "__main__":
    main()
