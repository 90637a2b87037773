"""TEST INFRASTRUCTURE: CPU oracle of the SGA hot path (see sga_oracle.py header).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
