"""CPU oracle for the ciphertext multiply + relinearize path.  TEST INFRASTRUCTURE ONLY (parity unpinned,
see oracle/model.py): importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never from alchemy_amd/."""
