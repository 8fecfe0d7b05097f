#!/usr/bin/env python3
"""Golden fixture for disco_amd/dnn/utils.py:normalization, produced by RUNNING THE REFERENCE'S OWN FUNCTION
(disco_theque/dnn/utils.py:14-41, exec'd from its source segment: the module itself cannot be imported, it pulls in the
training stack).  Writes tests/golden/dnn_normalization_ref.npz (build container only)."""
import ast
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/disco_theque/dnn/utils.py'


def main():
    src = open(REF).read()
    fn = next(ast.get_source_segment(src, n) for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'normalization')
    ns = {'torch': torch}
    exec(fn, ns)
    normalization = ns['normalization']
    rng = np.random.default_rng(77)
    d = {}
    x = rng.standard_normal((5, 21, 33)).astype(np.float32) * 3.0 + 0.5
    d['x'] = x
    for nt in ('scale_to_unit_norm', 'scale_to_1', 'center_and_scale', 'none'):
        for axis in (0, 1, 2):
            d[f'{nt}_axis{axis}'] = normalization(torch.from_numpy(x), None if nt == 'none' else nt, axis).numpy()
    np.savez_compressed(os.path.join(HERE, 'dnn_normalization_ref.npz'), **d)
    print('wrote dnn_normalization_ref.npz', len(d), 'arrays')


if __name__ == '__main__':
    main()
