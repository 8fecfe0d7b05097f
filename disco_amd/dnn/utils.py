"""The two helpers of disco_theque/dnn/utils.py that sit on the enhancement path: `tf_mask` (:44-71, a verbatim duplicate of
sigproc_utils.tf_mask in the reference, one implementation here) and the tensor `normalization` (:14-41).  The rest of that
module (data lists, training / evaluation loops) belongs to CRNN training and is out of scope."""
import torch

from ..sigproc_utils import tf_mask  # noqa: F401


def normalization(x, norm_type=None, axis=0):
    """Normalise tensor `x` along `axis` (dnn/utils.py:14-41): 'scale_to_unit_norm' (divide by the 2-norm), 'scale_to_1'
    (divide by the maximum), 'center_and_scale' (subtract the mean, divide by the unbiased standard deviation -- torch.std's
    default, as the reference calls it); anything else returns `x` unchanged, as the reference does.  Stays on x's device."""
    if norm_type == 'scale_to_unit_norm':
        return x / torch.linalg.vector_norm(x, dim=axis, keepdim=True)
    if norm_type == 'scale_to_1':
        return x / torch.amax(x, dim=axis, keepdim=True)
    if norm_type == 'center_and_scale':
        x = x - torch.mean(x, dim=axis, keepdim=True)
        return x / torch.std(x, dim=axis, keepdim=True)
    return x
