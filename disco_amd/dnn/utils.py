"""The reference keeps a verbatim duplicate of `tf_mask` here (disco_theque/dnn/utils.py:44-71)."""
from ..sigproc_utils import tf_mask  # noqa: F401
