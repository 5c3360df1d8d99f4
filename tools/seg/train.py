#!/usr/bin/env python3
"""`tools/seg/train.py` -- the spelling BASELINE.json uses for the segmentation training driver (upstream's is recalled as `train_seg.py`,
SURVEY.md Appendix B): both spellings run the same main()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_seg import main  # noqa: E402

if __name__ == "__main__":
    main()
