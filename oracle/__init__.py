"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference hot path (see erl_literal.py,
bwt_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (bwt_amd/) must never do so."""
