"""Step-1-only variant (disco_theque/speech_enhancement/get_z_signals.py:213-317): returns
(z_y, z_s, z_n, zn, masks_z) as lists over nodes of (F, T) arrays."""
from .tango import offline_tango as _two_step


def offline_tango(y, s, n, vads='irm1', mods=None, mask_for_z='local', z_sigs='zs_hat'):
    res = _two_step(y, s, n, vads=vads, mods=mods, mask_for_z=mask_for_z, z_sigs=z_sigs)
    return res[3], res[4], res[5], res[6], res[7]
