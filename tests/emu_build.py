"""Build the hipemu TEST library: the kernel sources of disco_amd/csrc compiled by g++ against
tests/hipemu (a CPU emulator of the HIP execution model).  Used ONLY by `-m "not gpu"` tests to check kernel
logic at toy sizes without a GPU; never loaded by the disco_amd package (which has no CPU path)."""
import ctypes
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, 'disco_amd', 'csrc')
# DISCO_CXXFLAGS (the -D switches of an A/B build, as disco_amd/build.py reads them) select a variant: its own objects and library
_EXTRA = os.environ.get('DISCO_CXXFLAGS', '').split()
_TAG = ('.' + hashlib.sha1(' '.join(_EXTRA).encode()).hexdigest()[:8]) if _EXTRA else ''
OUT = os.path.join(HERE, '_emu', f'libdisco_hipemu_TESTONLY{_TAG}.so')
OBJ = os.path.join(HERE, '_emu', 'obj' + _TAG)


def build_emu():
    units = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith('api_') and f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs += [os.path.join(HERE, 'hipemu', 'include', 'hip', 'hip_runtime.h'), os.path.join(REPO, 'include', 'disco_hip.h')]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in units + hdrs):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    flags = ['g++', '-O2', '-std=c++17', '-x', 'c++', '-fPIC', '-pthread', '-ffp-contract=off', '-I', os.path.join(HERE, 'hipemu', 'include')] + _EXTRA

    def compile_unit(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        if not (os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs)):
            subprocess.check_call(flags + ['-c', '-o', obj, src])
        return obj

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        objs = list(ex.map(compile_unit, units))
    # a sanitizer build (DISCO_CXXFLAGS='-fsanitize=address,undefined ...', tests/run_sanitized.sh) links its runtimes too
    subprocess.check_call(['g++', '-shared', '-pthread', '-o', OUT] + [f for f in _EXTRA if f.startswith('-fsanitize')] + objs)
    return OUT


def load_emu():
    from disco_amd import _lib
    return _lib.bind(ctypes.CDLL(build_emu()))


if __name__ == '__main__':
    print(build_emu())
