"""placeholder — replaced by the ctypes mirror of include/icicle_snark_hip.h"""
