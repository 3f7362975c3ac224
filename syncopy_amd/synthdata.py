"""Synthetic AnalogData generators with the reference's sampling scheme
(syncopy/synthdata/analog.py:20-48,186-252 and the per-trial seeding of
syncopy/synthdata/utils.py:53-55), used for parity fixtures and the benchmark inputs."""
import numpy as np

from .datatype import AnalogData


def _trial_seeds(seed, nTrials):
    if seed is None:
        return [None] * nTrials
    return list(np.random.default_rng(seed).integers(1_000_000, size=nTrials))


def _ar2_trial(AdjMat, nSamples, alphas, seed):
    AdjMat = np.asarray(AdjMat).astype(np.float32)
    nChannels = AdjMat.shape[0]
    alpha1, alpha2 = alphas
    M = np.diag(nChannels * [alpha1]) + AdjMat.T
    sig = np.zeros((nSamples, nChannels), dtype=np.float32)
    rng = np.random.default_rng(seed)
    sig[:2, :] = rng.normal(size=(2, nChannels))
    for i in range(2, nSamples):
        sig[i, :] = M @ sig[i - 1, :] + alpha2 * sig[i - 2, :]
        sig[i, :] += rng.normal(size=nChannels)
    return sig


def ar2_network(AdjMat=None, nSamples=1000, alphas=(0.55, -0.8), seed=None, nTrials=100, samplerate=1000):
    """Network of coupled AR(2) processes; entry (i, j) of `AdjMat` couples i -> j."""
    if AdjMat is None:
        AdjMat = np.zeros((2, 2), dtype=np.float32)
        AdjMat[1, 0] = 0.25
    if nTrials is None:
        return _ar2_trial(AdjMat, nSamples, alphas, seed)
    trials = [_ar2_trial(AdjMat, nSamples, alphas, s) for s in _trial_seeds(seed, nTrials)]
    return _collect(trials, samplerate)


def white_noise(nSamples=1000, nChannels=2, seed=None, nTrials=100, samplerate=1000):
    def one(s):
        return np.random.default_rng(s).normal(size=(nSamples, nChannels)).astype("f4")
    if nTrials is None:
        return one(seed)
    return _collect([one(s) for s in _trial_seeds(seed, nTrials)], samplerate)


def _collect(trials, samplerate):
    """Trials stacked along time; like the reference's generator-built AnalogData every trial
    gets a trigger offset of -1 s (time axis starts at -1.0)."""
    n = trials[0].shape[0]
    starts = np.arange(len(trials)) * n
    trl = np.stack([starts, starts + n, np.full(len(trials), -samplerate)], axis=1)
    return AnalogData(np.concatenate(trials, axis=0), samplerate=samplerate, trialdefinition=trl)


def ar2_uncoupled_fast(nChannels, nSamples, nTrials, alphas=(0.55, -0.8), seed=0, samplerate=1000, device=None):
    """Uncoupled AR(2) noise for large benchmark inputs, generated on the GPU (same process
    parameters as ar2_network with AdjMat = 0; not the reference's random stream)."""
    import torch
    dev = torch.device("cuda" if device is None else device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    a1, a2 = alphas
    x = torch.empty((nTrials, nSamples, nChannels), dtype=torch.float32, device=dev)
    x[:, :2] = torch.randn((nTrials, 2, nChannels), generator=g, device=dev)
    for i in range(2, nSamples):
        x[:, i] = a1 * x[:, i - 1] + a2 * x[:, i - 2] + torch.randn((nTrials, nChannels), generator=g, device=dev)
    return x.reshape(nTrials * nSamples, nChannels)
