"""CPU oracle for the HGI hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  rustyhgi_amd never imports this package.
"""
