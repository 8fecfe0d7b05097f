"""On-disk layout of the compressed-signal ("z") dataset the reference writes after step 1
(disco_theque/speech_enhancement/get_z_signals.py:320-359), for batches of rooms processed on the GPU.

    <root>/raw/<snr-dir>/zs_hat/<i_rir>_<noise>_Node-<k+1>.npy            z_y   (F, T) complex64   (get_z_signals.py:350-351)
    <root>/raw/<snr-dir>/zn_hat/...                                       zn    (F, T) complex64   (:352-353)
    <root>/normed/abs/<snr-dir>/zs_hat/...                                |z_y| (F, T) float32     (:354-355)
    <root>/normed/abs/<snr-dir>/zn_hat/...                                |zn|  (F, T) float32     (:356-357)

`<snr-dir>` = get_directory_name(snr_range) (get_z_signals.py:37-41), e.g. '0-6'.  A room is skipped when its LAST node's
normed/abs/zn_hat file already exists, as the reference does (:329-332) -- note that the reference tests the name WITHOUT
the '.npy' suffix np.save appends, so its check never fires; here the real file name is tested.
The engine's arrays are frame-major (R, K, T, F); files hold the reference's (F, T).
"""
import os

import numpy as np

SUBDIRS = (('raw', 'zs_hat'), ('raw', 'zn_hat'), (os.path.join('normed', 'abs'), 'zs_hat'), (os.path.join('normed', 'abs'), 'zn_hat'))


def get_directory_name(snr_range):
    """get_z_signals.py:37-41 -- [[0, 6]] -> '0-6', [[3, 6], [5, 15]] -> '3-6_5-15'."""
    return '_'.join('{}-{}'.format(str(r[0]), str(r[1])) for r in snr_range)


def _path(root, top, dirry, kind, i_rir, noise, node):
    return os.path.join(root, top, dirry, kind, '{}_{}_Node-{}.npy'.format(str(i_rir), noise, str(node + 1)))


def already_processed(root, i_rir, noise, nb_nodes, snr_range=((0, 6),)):
    return os.path.isfile(_path(root, os.path.join('normed', 'abs'), get_directory_name(snr_range), 'zn_hat', i_rir, noise,
                                nb_nodes - 1))


def write_z_dataset(root, rir_ids, noise, z_y, zn, snr_range=((0, 6),), skip_existing=True):
    """z_y, zn: (R, K, T, F) complex arrays (NumPy, or anything with .numpy(): DevBuf / CPU torch tensors) from
    `offline_tango_batched` / `Engine`; rir_ids: the R room identifiers.  Returns the list of rooms written."""
    z_y = np.asarray(z_y.numpy() if hasattr(z_y, 'numpy') else z_y)
    zn = np.asarray(zn.numpy() if hasattr(zn, 'numpy') else zn)
    assert z_y.shape == zn.shape and z_y.ndim == 4 and len(rir_ids) == z_y.shape[0]
    R, K = z_y.shape[:2]
    dirry = get_directory_name(snr_range)
    for top, kind in SUBDIRS:
        os.makedirs(os.path.join(root, top, dirry, kind), exist_ok=True)
    written = []
    for r, i_rir in enumerate(rir_ids):
        if skip_existing and already_processed(root, i_rir, noise, K, snr_range):
            continue
        for k in range(K):
            a = np.ascontiguousarray(z_y[r, k].T.astype(np.complex64))
            b = np.ascontiguousarray(zn[r, k].T.astype(np.complex64))
            np.save(_path(root, 'raw', dirry, 'zs_hat', i_rir, noise, k), a)
            np.save(_path(root, 'raw', dirry, 'zn_hat', i_rir, noise, k), b)
            np.save(_path(root, os.path.join('normed', 'abs'), dirry, 'zs_hat', i_rir, noise, k), np.abs(a))
            np.save(_path(root, os.path.join('normed', 'abs'), dirry, 'zn_hat', i_rir, noise, k), np.abs(b))
        written.append(i_rir)
    return written


def read_z(root, i_rir, noise, node, kind='zs_hat', normed=False, snr_range=((0, 6),)):
    """One file back, as the training loaders of the reference read them: (F, T)."""
    top = os.path.join('normed', 'abs') if normed else 'raw'
    return np.load(_path(root, top, get_directory_name(snr_range), kind, i_rir, noise, node))
