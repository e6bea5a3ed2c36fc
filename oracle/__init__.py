"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/if_oracle.c header).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from imageflow_amd/.
"""
