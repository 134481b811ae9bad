"""CPU oracle for the go_zero hot path. TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from alphazero_amd/ or go_zero/."""
