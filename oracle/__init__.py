"""ORACLE package -- test infrastructure only (see oracle/oracle.cpp header).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by glaze_amd.
"""
