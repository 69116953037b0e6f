"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference's hot path; may be
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
hironaka_amd/."""
