"""MI355X-native train-step path of the polyphonic chord/texture disentanglement VAE.

Python host mirroring the reference's class surface (`model`, `ptvae`, `amc_dl.torch_plus`,
`dataset_loaders`) over `csrc/libptvae_hip.so` (C ABI in `include/ptvae_hip.h`).  See DESIGN.md."""
import torch as _torch

# encoder branches run (and therefore back-propagate) on sibling HIP streams by design
if hasattr(_torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
    _torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
