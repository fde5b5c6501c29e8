"""`HYGNN` is the extension's earlier name, still used by the reference driver
(HC-SpMM_main.py:52 calls HYGNN.preprocess while importing HCSPMM).  Same module under both names."""
from HCSPMM import *  # noqa: F401,F403
from HCSPMM import preprocess  # noqa: F401
