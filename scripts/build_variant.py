"""Builds an experimental variant of libpglamd.so next to the product one: extra -D macros, own object directory.
    python scripts/build_variant.py NAME MACRO[=VALUE] ...   ->  pgl_amd/csrc/variants/libpglamd_NAME.so
Run anything against it with PGLAMD_LIB=<that path> (same C ABI; the product build is untouched)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgl_amd import _build

name, defines = sys.argv[1], sys.argv[2:]
vdir = os.path.join(_build.CSRC, "variants")
os.makedirs(vdir, exist_ok=True)
print(_build.build(force=False, verbose=False, defines=defines, lib=os.path.join(vdir, "libpglamd_%s.so" % name),
                   obj=os.path.join(_build.CSRC, "build", "variant_" + name)))
