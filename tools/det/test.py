#!/usr/bin/env python3
"""`tools/det/test.py` -- the spelling BASELINE.json uses for the detection evaluation driver.  Upstream's script is recalled as
`test_codet.py` (`/root/reference/README.md:101` names only the directory): both spellings run the same main()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_codet import main  # noqa: E402

if __name__ == "__main__":
    main()
