#!/usr/bin/env python3
"""`tools/det/train.py` -- the spelling BASELINE.json uses for the detection training driver.  Upstream's script is recalled as
`train_codet.py` (`/root/reference/README.md:101` names only the directory): both spellings run the same main()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_codet import main  # noqa: E402

if __name__ == "__main__":
    main()
