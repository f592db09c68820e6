"""CPU oracle for the U-Net hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (Jarvis73/BoxSegLiver) ships no tests, golden
vectors or checkpoints for this path, and its arithmetic lives in the absent
third-party dependency tensorflow-gpu==1.13 (requirements.txt:2), which cannot
be installed or imported here.  This package restates the *published* TF-1.13 /
tf.contrib.slim semantics the reference relies on and
the explicit formulas in the reference's loss_metrics.py, and is pinned only by
analytic known-answer tests written by this repo (tests/test_oracle_kat.py)
plus an independent pure-numpy loop restatement (oracle/naive.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (boxsegliver_amd/) never does.
"""
