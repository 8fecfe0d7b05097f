"""CPU oracle for the DISCO / Tango MWF hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``disco_amd``) may
import this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Parity pinning (see oracle/PINNING.md):
  * ``intern_filter``, ``tf_mask`` and the whole ``offline_tango`` loop nest are
    pinned against outputs of the REFERENCE'S OWN CODE, executed in the build
    container by ``tests/golden/make_golden.py`` (the function bodies are
    loaded from /root/reference at run time, never copied) and committed as
    fixtures under ``tests/golden/``.
  * The STFT/iSTFT are third-party (librosa, un-vendored, version unpinned in
    the reference): restated here from the published algorithm and
    cross-checked against two independent implementations (``torch.stft`` and
    ``scipy.signal.stft``).  The reference has no test, fixture or golden
    vector for them: that part is "parity unpinned" by the reference itself.
"""
