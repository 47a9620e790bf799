"""
scroll-prover_amd -- MI355X-native BN254 G1-MSM / Fr-NTT hot path behind scroll-prover's halo2_proofs seam.

The product is libmi355zk.so (csrc/*.hip, C-ABI in include/mi355zk.h).  This package is the thin Python
host side: a ctypes binding (`_capi`) plus `halo2`, a mirror of the halo2_proofs operator surface
(best_multiexp, best_fft, EvaluationDomain, ParamsKZG) so that tests and the bench read like the
reference's own call sites.  The directory name contains '-', so import it through
`__graft_entry__.load_package()` (importlib), which registers it as `scroll_prover_amd`.

There is NO CPU fallback: importing works anywhere (so the symbol table can be checked without a GPU),
but every compute call raises unless libmi355zk.so is present AND mi355_init() bound a gfx950 device.
"""
from . import _capi  # noqa: F401
from ._capi import Mi355Error, lib, init, shutdown  # noqa: F401
from . import halo2  # noqa: F401
from . import distributed  # noqa: F401
from . import protocols  # noqa: F401
from . import replay  # noqa: F401
