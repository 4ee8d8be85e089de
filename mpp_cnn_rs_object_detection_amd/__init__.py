"""MI355X-native MPP / RJMCMC sampling path (drop-in for the reference's ``models/mpp`` inference path).

Host code in Python mirrors the reference's interfaces; the work happens in ``libmppgpu.so``
(hand-written HIP for gfx950, C ABI in ``include/mpp_hip.h``).  PyTorch is used only for the two
U-Nets and for ``torch.distributed``.
"""
__version__ = "0.1.0"
