"""Build the hipemu TEST library: the kernel sources of disco_amd/csrc compiled by g++ against
tests/hipemu (a CPU emulator of the HIP execution model).  Used ONLY by `-m "not gpu"` tests to check kernel
logic at toy sizes without a GPU; never loaded by the disco_amd package (which has no CPU path)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(REPO, 'disco_amd', 'csrc', 'disco_hip.hip')
OUT = os.path.join(HERE, '_emu', 'libdisco_hipemu_TESTONLY.so')


def build_emu():
    deps = [os.path.join(REPO, 'disco_amd', 'csrc', f) for f in os.listdir(os.path.join(REPO, 'disco_amd', 'csrc'))]
    deps += [os.path.join(HERE, 'hipemu', 'include', 'hip', 'hip_runtime.h'), os.path.join(REPO, 'include', 'disco_hip.h')]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ['g++', '-O2', '-std=c++17', '-x', 'c++', '-fPIC', '-shared', '-pthread', '-ffp-contract=off',
           '-I', os.path.join(HERE, 'hipemu', 'include'), '-o', OUT, SRC]
    subprocess.check_call(cmd)
    return OUT


def load_emu():
    from disco_amd import _lib
    return _lib.bind(ctypes.CDLL(build_emu()))


if __name__ == '__main__':
    print(build_emu())
