"""Step-1-only variant (disco_theque/speech_enhancement/get_z_signals.py:213-317): returns
(z_y, z_s, z_n, zn, masks_z) as lists over nodes of (F, T) arrays.  Runs step 1 ONLY (disco_tango_reference with
steps = 1): no step-2 statistics, no step-2 constraints -- `mods` may hold the single step-1 model the reference reads
(mods[0], get_z_signals.py:279), and mask_for_z only matters through the oracle-statistics switch (get_z_signals.py:283)."""
import numpy as np

from .tango import _as_batch, _offline_tango_ragged, offline_tango_batched


def offline_tango(y, s, n, vads='irm1', mods=None, mask_for_z='local'):
    names = ['z_y', 'z_s', 'z_n', 'zn', 'masks_z']
    if isinstance(vads, (list, tuple)):
        vads = vads[0]                                        # only the step-1 mask type is used (get_z_signals.py:277-281)
    yb, sb, nb = _as_batch(y), _as_batch(s), _as_batch(n)
    if yb is None or sb is None or nb is None:                # ragged channel counts: the staged per-node path
        d = _offline_tango_ragged(y, s, n, vads, mask_for_z, steps=1)
        return tuple([np.ascontiguousarray(v.T) for v in d[nm]] for nm in names)
    mods = None if mods is None else [mods[0] if isinstance(mods, (list, tuple)) else mods, None]
    d = offline_tango_batched(yb, sb, nb, vads=[vads, vads], mods=mods, mask_for_z=mask_for_z, steps=1)
    K = d['z_y'].shape[1]
    return tuple([np.ascontiguousarray(d[nm][0, k].T) for k in range(K)] for nm in names)
