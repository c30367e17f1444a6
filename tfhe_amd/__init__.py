"""MI355X-native engine for the GSW hot path of n0900/tfhe (C-ABI HIP library + host mirror)."""
from ._lib import FLAG_NAND, GswError, LIB_PATH  # noqa: F401
from .gsw import (GSW, Ciphertext, DeviceBuffer, Engine, Fp, PreparedCache, PublicKey, Ring, SecretKey,  # noqa: F401
                  Zpow2)
