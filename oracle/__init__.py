"""CPU oracle -- test infrastructure only.  PARITY UNPINNED: the
reference tree holds no code to pin it against.  Never imported by the product package."""
