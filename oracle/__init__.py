"""oracle/ -- CPU restatements of the reference's arithmetic.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by trackformer_amd/."""
