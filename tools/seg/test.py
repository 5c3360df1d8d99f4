#!/usr/bin/env python3
"""`tools/seg/test.py` -- the spelling BASELINE.json uses for the segmentation evaluation driver (upstream's is recalled as `test_seg.py`,
SURVEY.md Appendix B): both spellings run the same main()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_seg import main  # noqa: E402

if __name__ == "__main__":
    main()
