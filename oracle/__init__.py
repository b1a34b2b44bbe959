"""ORACLE -- test infrastructure only (see the headers of the modules in here).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
