"""Test infrastructure: CPU restatements of the reference's PsiCMPS scan (see cmps_oracle.py header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
