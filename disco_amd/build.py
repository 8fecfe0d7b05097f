"""Build libdisco_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repository snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'disco_hip.hip')
OUT = os.path.join(HERE, 'lib', 'libdisco_hip.so')
DEPS = [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc'))] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'disco_hip.h')]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build_hip(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # -fno-slp-vectorize: hipcc's SLP pass fuses adjacent f32 ops into v_pk_*_f32, which issue slower than the scalar
    # pair on gfx950 (guide: "packed f32 VALU ... an anti-lever"); measured -0.9 ms on k_stft_cov, -0.6 ms on k_step2_cov_fused
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-shared', '-fPIC', '-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
