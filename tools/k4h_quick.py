import torch, numpy as np, time
from syncopy_amd import backend as be
torch.manual_seed(0)
R, F, C = 700, 600, 256
spec = (torch.randn(R, F, C, device="cuda") + 1j*torch.randn(R, F, C, device="cuda")).to(torch.complex64)
gain = torch.logspace(-12, 6, C, device="cuda")
spec = (spec * gain).contiguous()
acc = torch.zeros(F, C, C, dtype=torch.complex64, device="cuda")
be.csd_accumulate(spec, acc)
print("fallbacks", be.csd_split_fallbacks())
for f in (0, 1, 255, 256, 511, 512, 599):
    x = spec[:, f, :].to(torch.complex128)
    ref = x.T @ x.conj()
    d = torch.sqrt(torch.outer(ref.diagonal().real, ref.diagonal().real))
    tril = torch.tril(torch.ones(C, C, dtype=torch.bool, device="cuda"))
    err = ((acc[f].to(torch.complex128) - ref).abs() / d)[tril].max().item()
    print(f, "err", err)
# absmax supplied
am = spec.reshape(-1, C).real.abs().amax(0).maximum(spec.reshape(-1, C).imag.abs().amax(0)).float().contiguous()
acc2 = torch.zeros_like(acc)
be.csd_accumulate(spec, acc2, absmax=am)
print("same with supplied absmax:", torch.equal(acc, acc2), be.csd_split_fallbacks())
# NaN / Inf propagate
spec2 = spec.clone(); spec2[5, 7, 9] = float("nan"); spec2[6, 300, 11] = float("inf")
acc3 = torch.zeros_like(acc)
be.csd_accumulate(spec2, acc3)
print("fallbacks with nan/inf:", be.csd_split_fallbacks(), "nan row in f=7:", torch.isnan(acc3[7, 9, :10].real).all().item(), "others finite:", torch.isfinite(acc3[8].real).all().item())
import os
