"""MI355X-native engine for the go_zero self-play hot path (rules, batched PUCT search, network
inference). Host side in Python, compute in libazamd.so (HIP, gfx950)."""
__all__ = ["engine", "_native"]
