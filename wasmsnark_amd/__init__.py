"""wasmsnark_amd -- MI355X-native BN128 Groth16 prove hot path (drop-in for iden3/wasmsnark's
groth16GenProof / genZKSnarkProof).  The compute lives in libwsnark.so (hand-written HIP, C ABI in
include/wsnark.h); this package is the host-side mirror of the reference's JS API."""
from .bn128 import Bn128, ProvingKey, build, genZKSnarkProof, groth16GenProof, proof_from_bytes  # noqa: F401
from ._lib import WsnarkError, load  # noqa: F401

buildBn128 = build  # index.js:21 export name
