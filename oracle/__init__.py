"""ORACLE package -- test infrastructure.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; marbler_amd/ never does."""
