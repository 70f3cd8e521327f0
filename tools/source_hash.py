"""sha256 over the sources that define the compiled kernels (csrc headers, the per-target instantiation units and the
Makefile's flags): the identity of a build as far as kernel SPEED is concerned.  csrc/Makefile compiles it into the library
(ptrwm_source_hash()); tools/form_sweep.py writes it into the sweep the AUTO form rule is fitted on and tools/form_fit.py
copies it into csrc/form_table.inc (ptrwm_form_table_source_hash()): a table fitted on other kernels than the ones in the
library is then visible (tests/test_capi_library.py warns), not silent.

    python tools/source_hash.py        prints the 64 hex digits
"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rwm-pt-pytorch_amd", "csrc")
# (capi.hip and form_table.inc are not part of it: the dispatch rule and the table are what the hash is compared WITH)
FILES = ["kernel.h", "quad.h", "philox.h", "proposals.h", "targets.h", "variants.h", "Makefile"]


def source_hash() -> str:
    h = hashlib.sha256()
    names = FILES + sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "variants_*.hip")) + glob.glob(os.path.join(CSRC, "quad_*.hip")))
    for n in names:
        h.update(n.encode() + b"\0")
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
