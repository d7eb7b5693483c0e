"""Build an experimental variant of libelvis_amd.so: tools/variants/conv_variant.hip (= conv.hip + the timing-only hooks of
tools/variants/conv_hooks.h) with extra -D flags:
   python tools/build_variant.py NAME -DELVIS_EXP_X ...   ->  elvis_amd/lib/variants/NAME.so
Used for same-box A/B timing (swap the .so on the GPU box); never shipped."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elvis_amd import _build as B

name, extra = sys.argv[1], sys.argv[2:]
B.build(verbose=False)
vdir = os.path.join(B.LIBDIR, "variants"); os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, name + "_conv.o")
subprocess.check_call([B._hipcc()] + B.FLAGS + extra + ["-I", B.CSRC, "-c", os.path.join(ROOT, "tools", "variants", "conv_variant.hip"), "-o", obj])
objs = [os.path.join(B.OBJDIR, s.replace(".hip", ".o")) for s in B.SOURCES if s != "conv.hip"] + [obj]
out = os.path.join(vdir, name + ".so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", out] + objs)
os.remove(obj)
print(out)
