"""Host-side mirror of the reference's `models` package for the hot path:
same module paths, class / factory names, constructor arguments, attributes and state-dict
keys; every forward/backward runs on the hand-written gfx950 kernels of libppt_hip.so."""
