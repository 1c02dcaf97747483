import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


# moduli used across the suite ---------------------------------------------------------------------
# reference example moduli (examples/Arithmetic.hs:31-34): all = 1 mod 512
ARITH_QS = [268440577, 8392193, 1073750017]
# the four largest primes < 2^31 that are 1 mod 2^16
CFG3_QS = [2147352577, 2146959361, 2146041857, 2145976321]
# the six largest primes < 2^30 that are 1 mod 2^17: 4q fits a 32-bit word, the library runs Harvey's lazy butterflies on such rings
# (the size of the reference's Tunnel.hs moduli and of HomomRLWR's rounding moduli, examples/Tunnel.hs:34-38, HomomRLWR.hs:38-40)
Q30_QS = [1073479681, 1071513601, 1070727169, 1068236801, 1065484289, 1064697857]
# config 2: largest prime < 2^60 that is 1 mod 2^15
CFG2_Q60 = 1152921504606748673


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import cref
    cref.build()
    return cref
