# -*- coding: utf-8 -*-
"""
Golden-vector generator: runs the REAL reference (esi-neuroscience/syncopy,
mounted read-only at /root/reference) in the build container and writes small
input/output fixtures to tests/golden/*.npz.

Run through ``oracle/make_golden.sh`` (it sets up the interpreter, SPYDIR and
the two empty stand-in modules for the optional cluster / curve-fitting
imports ``dask_jobqueue`` and ``fooof`` that the reference imports at module
load but never touches on this path).  The reference never ships: only the
arrays written here do.

TEST INFRASTRUCTURE ONLY - see oracle/spy_oracle.py header.
"""
import hashlib
import os
import sys

import numpy as np

import syncopy as spy
from syncopy import synthdata
from syncopy.connectivity.wilson_sf import wilson_sf, regularize_csd
from syncopy.connectivity.granger import granger
from syncopy.connectivity.csd import csd as ref_csd
from syncopy.specest.mtmfft import mtmfft as ref_mtmfft
from syncopy.specest.mtmconvol import mtmconvol as ref_mtmconvol
from syncopy.specest.wavelet import wavelet as ref_wavelet
from syncopy.specest.wavelets import Morlet

OUT = sys.argv[1] if len(sys.argv) > 1 else "tests/golden"
ONLY = set(sys.argv[2:])          # optional: names of the fixture files to (re)write, default all
os.makedirs(OUT, exist_ok=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def trials_of(d):
    return [np.array(t) for t in d.trials]


def ca(data, sl=slice(None), **opts):
    r = spy.connectivityanalysis(data, **opts)  # keep the object alive while reading its HDF5 file
    return np.array(r.data[sl])


def save(name, **kw):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


# ---------------------------------------------------------------- c1 (BASELINE config 1)
c1 = synthdata.ar2_network(nTrials=20, nSamples=2000, AdjMat=np.zeros((16, 16)), seed=42)
s = spy.freqanalysis(c1, method="mtmfft", tapsmofrq=2)
coh = spy.connectivityanalysis(c1, method="coh", tapsmofrq=2)
csd_lim = spy.connectivityanalysis(c1, method="csd", tapsmofrq=2, foilim=[0, 60])
save(
    "c1",
    data_sha256=sha(c1.data[()]),
    trial0=np.array(c1.trials[0]),
    trial19_tail=np.array(c1.trials[19])[-4:],
    sampleinfo=c1.sampleinfo,
    samplerate=c1.samplerate,
    pow=s.data[()],
    freq=s.freq,
    taper=np.array([str(t) for t in s.taper]),
    coh_abs=coh.data[()],
    csd_foilim_0_60=csd_lim.data[()],
    csd_freq=csd_lim.freq,
)

# ---------------------------------------------------------------- small coupled network: csd / coh / granger
adj = np.zeros((5, 5))
adj[0, 1] = 0.25
adj[3, 2] = 0.2
adj[1, 4] = 0.15
n5 = synthdata.ar2_network(nTrials=60, nSamples=1000, AdjMat=adj, seed=7, samplerate=200)
kw = {}
for outp in ["abs", "pow", "complex", "angle", "imag", "real"]:
    kw["coh_" + outp] = ca(n5, method="coh", tapsmofrq=3, output=outp)
kw["csd"] = ca(n5, method="csd", tapsmofrq=3)
kw["csd_keeptrials_first3"] = ca(n5, slice(0, 3), method="csd", tapsmofrq=3, keeptrials=True)
kw["coh_hann_pad"] = ca(n5, method="coh", taper="hann", pad="nextpow2")
kw["coh_foi"] = ca(n5, method="coh", tapsmofrq=3, foi=[10, 20.2, 40, 40.1, 77])
g = spy.connectivityanalysis(n5, method="granger", tapsmofrq=3)
kw["granger"] = g.data[()]
kw["granger_info"] = np.array(
    [float(g.info["converged"]), g.info["max rel. err"], g.info["reg. factor"], g.info["initial cond. num"]]
)
kw["freq"] = g.freq
save("conn5", adj=adj, data=np.stack(trials_of(n5)), samplerate=n5.samplerate, **kw)

# ---------------------------------------------------------------- jackknife (connectivity_analysis.py:601-606,736-757; jackknifing.py)
kw = {}
n5j = n5.selectdata(trials=np.arange(20))          # 20 trials: 20 leave-one-out replicates


def jk(name, **opts):
    r = spy.connectivityanalysis(n5j, jackknife=True, **opts)
    kw[name] = r.data[()]
    kw[name + "_jack_var"] = np.array(r.jack_var)
    kw[name + "_jack_bias"] = np.array(r.jack_bias)


jk("coh_abs", method="coh", tapsmofrq=3)
jk("coh_complex", method="coh", tapsmofrq=3, output="complex", foilim=[5, 60])
jk("granger", method="granger", tapsmofrq=3)
save("jackknife", adj=adj, data=np.stack(trials_of(n5j)), samplerate=n5j.samplerate, **kw)

# ---------------------------------------------------------------- SpectralData-input connectivity (connectivity_analysis.py:475-538,
# ST_compRoutines.py:30-117): freqanalysis(output="fourier", keeptapers=True) chained into connectivityanalysis
kw = {}
spec5 = spy.freqanalysis(n5j, method="mtmfft", tapsmofrq=3, output="fourier", keeptapers=True, foilim=[0, 60])
kw["spec_freq"] = spec5.freq
kw["spec_first_trial"] = spec5.data[0:1]            # the full (20, 29, 301, 5) spectra are 7 MB: the tests chain their own
kw["chain_coh"] = ca(spec5, method="coh")
kw["chain_coh_complex"] = ca(spec5, method="coh", output="complex")
kw["chain_csd"] = ca(spec5, method="csd")
kw["chain_csd_keeptrials_first3"] = ca(spec5, slice(0, 3), method="csd", keeptrials=True)
spec5g = spy.freqanalysis(n5j, method="mtmfft", tapsmofrq=3, output="fourier", keeptapers=True, demean_taper=True)
gg = spy.connectivityanalysis(spec5g, method="granger")
kw["chain_granger"] = gg.data[()]
save("chain", data=np.stack(trials_of(n5j)), samplerate=n5j.samplerate, **kw)

# ---------------------------------------------------------------- channelcmb, ppc, corr (connectivity_analysis.py:335-381,501-529,
# 551-566,624-663,681-733,760-763; ST_compRoutines.py:159-233,466-584; AV_compRoutines.py:166-228)
kw = {}
for tag, cmb in (("idx", [[3, 0], [1, 2]]), ("str", [["channel2", "channel4"], ["channel4", "channel1"]])):
    for meth in ("coh", "csd", "ppc"):
        r = spy.connectivityanalysis(spec5, method=meth, channelcmb=cmb)
        kw[f"cmb_{tag}_{meth}"] = r.data[()]
        kw[f"cmb_{tag}_{meth}_channel_i"] = np.array(r.channel_i, dtype="U")
        kw[f"cmb_{tag}_{meth}_channel_j"] = np.array(r.channel_j, dtype="U")
    r = spy.connectivityanalysis(spec5g, method="granger", channelcmb=cmb)
    kw[f"cmb_{tag}_granger"] = r.data[()]
    kw[f"cmb_{tag}_granger_channel_i"] = np.array(r.channel_i, dtype="U")
    kw[f"cmb_{tag}_granger_channel_j"] = np.array(r.channel_j, dtype="U")
kw["ppc_spec"] = ca(spec5, method="ppc")
kw["ppc_analog"] = ca(n5j, method="ppc", tapsmofrq=3, foilim=[0, 60])
kw["ppc_analog_hann"] = ca(n5j, method="ppc", taper="hann", pad="nextpow2")
cc = spy.connectivityanalysis(n5j, method="corr")
kw["corr"] = cc.data[()]
kw["corr_time"] = np.array(cc.time[0])
kw["corr_poly1"] = ca(n5j, method="corr", polyremoval=1)
kw["corr_keeptrials_first3"] = ca(n5j, slice(0, 3 * 500), method="corr", keeptrials=True)
n5odd = n5j.selectdata(latency=[-1, 3.99 - 1])            # odd number of samples per trial (lags: nSamples // 2 + 1)
kw["corr_odd_nsamples"] = np.array([np.diff(n5odd.sampleinfo)[0, 0]])
kw["corr_odd"] = ca(n5odd, method="corr")
save("conn_next", data=np.stack(trials_of(n5j)), samplerate=n5j.samplerate, **kw)

# ---------------------------------------------------------------- mtmfft option sweep (incl. unequal trial lengths + selections)
rng = np.random.default_rng(2024)
lens = [1500, 2000, 1800, 2000]
trl = []
tt = np.arange(2000) / 1000.0
for n in lens:
    x = rng.normal(size=(n, 4)).astype("f4")
    x[:, 0] += 3 * np.cos(2 * np.pi * 30 * tt[:n]).astype("f4")
    x[:, 2] += (2 * np.cos(2 * np.pi * 111 * tt[:n]) + 5 + 0.002 * np.arange(n)).astype("f4")
    trl.append(x)
# trials live in one (time x channel) block; the third one overlaps the second by 100 samples
# and every trial starts 250 samples before its trigger (offset -250)
block = np.concatenate(trl, axis=0)
starts = np.array([0, 1500, 3400, 5300])
trldef = np.stack([starts, starts + np.array(lens), np.full(4, -250)], axis=1)
uneq = spy.AnalogData(data=block, samplerate=1000, trialdefinition=trldef)
trl = [np.array(t) for t in uneq.trials]
kw = {"block": block, "trialdefinition": uneq.trialdefinition, "samplerate": uneq.samplerate}


def fa(name, data=uneq, **opts):
    r = spy.freqanalysis(data, method="mtmfft", **opts)
    kw[name] = r.data[()]
    kw[name + "_freq"] = r.freq
    kw[name + "_trialdef"] = r.trialdefinition


fa("v_fourier_keeptapers", tapsmofrq=3, keeptapers=True, output="fourier")
fa("v_hann_nextpow2", taper="hann", pad="nextpow2", output="pow")
fa("v_foilim_linear_abs", taper="hann", foilim=[10, 100], polyremoval=1, output="abs")
fa("v_foi_boxcar_avg", taper=None, foi=[5, 30.3, 30.4, 111, 250], keeptrials=False, polyremoval=0)
fa("v_pad3s_dpss", tapsmofrq=2, pad=3.0, output="pow")
fa("v_ftcompat", taper="hann", ft_compat=True, pad="nextpow2")
fa("v_demean_taper", tapsmofrq=4, demean_taper=True, keeptapers=True, output="fourier", polyremoval=None)
fa("v_kaiser", taper="kaiser", taper_opt={"beta": 4.5}, output="real")
fa("v_ntaper3", tapsmofrq=4, nTaper=3, output="pow")
fa("v_select", tapsmofrq=2, select={"trials": [2, 0, 3], "channel": [3, 1], "latency": [0.1, 1.2]})
for outp in ["imag", "angle", "absreal", "absimag"]:
    fa("v_out_" + outp, taper="hann", output=outp, select={"trials": [1]})
save("mtmfft_variants", **kw)

# ---------------------------------------------------------------- time-frequency: mtmconvol + wavelet
tf = synthdata.ar2_network(nTrials=3, nSamples=2000, AdjMat=np.zeros((4, 4)), seed=11)
kw = {"data": np.stack(trials_of(tf)), "samplerate": tf.samplerate, "trialdefinition": tf.trialdefinition}


def tfa(name, **opts):
    r = spy.freqanalysis(tf, **opts)
    kw[name] = r.data[()]
    kw[name + "_freq"] = r.freq
    kw[name + "_trialdef"] = r.trialdefinition
    kw[name + "_time0"] = np.array(r.time[0])


tfa("conv_hann_half", method="mtmconvol", taper="hann", t_ftimwin=0.5, toi=0.5)
tfa("conv_hann_pow2", method="mtmconvol", taper="hann", t_ftimwin=0.256, toi=0.75, foilim=[0, 200])
tfa("conv_dpss_keep", method="mtmconvol", tapsmofrq=2, t_ftimwin=0.4, toi=0.5, keeptapers=True, output="fourier",
    foilim=[0, 120])
tfa("conv_all", method="mtmconvol", taper="hann", t_ftimwin=0.1, toi="all", foi=[20, 40, 60], polyremoval=1)
tfa("conv_toi_equi", method="mtmconvol", taper="hann", t_ftimwin=0.05, toi=np.arange(-0.5, 0.5, 0.01))
tfa("conv_toi_irreg", method="mtmconvol", taper="hann", t_ftimwin=0.3, toi=np.array([-0.6, -0.45, 0.0, 0.31]))
tfa("wav_all", method="wavelet", wavelet="Morlet", width=6, foi=np.arange(10, 110, 10), toi="all")
tfa("wav_toi", method="wavelet", wavelet="Morlet", width=4, foi=np.array([8.0, 33.0, 150.0]),
    toi=np.arange(-0.8, 0.8, 0.05), output="fourier")
tfa("wav_auto_scales", method="wavelet", wavelet="Morlet", toi="all", output="abs", keeptrials=False)
save("tf_variants", **kw)

# ---------------------------------------------------------------- the other wavelet functions (freqanalysis.py:55, wavelets.py:140-363)
kw = {"data": np.stack(trials_of(tf)), "samplerate": tf.samplerate, "trialdefinition": tf.trialdefinition}
tfa("paul4_fourier", method="wavelet", wavelet="Paul", order=4, foi=np.array([12.0, 40.0, 95.0]), toi="all", output="fourier")
tfa("paul6_toi_pow", method="wavelet", wavelet="Paul", order=6, foi=np.array([20.0, 60.0]), toi=np.arange(-0.6, 0.6, 0.02))
tfa("dog1_abs", method="wavelet", wavelet="DOG", order=1, foi=np.array([10.0, 30.0, 120.0]), toi="all", output="abs")
tfa("dog6_real_avg", method="wavelet", wavelet="DOG", order=6, foi=np.array([25.0, 80.0]), toi="all", output="real",
    keeptrials=False, polyremoval=1)
tfa("ricker_pow", method="wavelet", wavelet="Ricker", foi=np.array([15.0, 50.0, 150.0]), toi="all")
tfa("mexican_hat_auto", method="wavelet", wavelet="Mexican_hat", toi="all", output="abs", keeptrials=False)
tfa("paul4_auto", method="wavelet", wavelet="Paul", order=4, toi="all", keeptrials=False)
save("wavelet_families", **kw)

# ---------------------------------------------------------------- trial lengths behind the round-4 radix schedules
# (3 x a scheduled length through the radix-3 decimation, 10000 with split exchanges, 12000 = 6 x 2000 through HBM):
# mtmfft.py:80-129 takes any nSamples
kw = {}
for n in (100, 300, 400, 600, 768, 800, 1500, 2400, 3000, 3072, 4800, 6000, 8000, 10000, 12000):
    d = synthdata.ar2_network(nTrials=3, nSamples=n, AdjMat=np.zeros((4, 4)), seed=n)
    kw[f"n{n}_data"] = np.stack(trials_of(d))
    kw[f"n{n}_trialdefinition"] = d.trialdefinition
    r = spy.freqanalysis(d, method="mtmfft", tapsmofrq=2, keeptrials=False)
    kw[f"n{n}_pow_avg"] = r.data[()]
    kw[f"n{n}_freq"] = r.freq
    r = spy.freqanalysis(d, method="mtmfft", taper="hann", output="fourier", select={"trials": [1]})
    kw[f"n{n}_fourier_trial1"] = r.data[()]
    kw[f"n{n}_coh"] = ca(d, method="coh", tapsmofrq=2)
    r = spy.freqanalysis(d, method="mtmfft", taper="hann", polyremoval=1, foilim=[0, 100],
                         select={"latency": [-1.0, -1.0 + (n - 37) / 1000.0]}, pad=n / 1000.0)
    kw[f"n{n}_pow_pad"] = r.data[()]
save("lengths", **kw)

# ---------------------------------------------------------------- welch = mtmconvol + spy.mean(dim="time") (freqanalysis.py:1054-1056)
kw = {"data": np.stack(trials_of(tf)), "samplerate": tf.samplerate, "trialdefinition": tf.trialdefinition}
tfa("welch_hann_half", method="welch", taper="hann", t_ftimwin=0.5, toi=0.5)
tfa("welch_dpss_avg", method="welch", tapsmofrq=4, t_ftimwin=0.4, toi=0.25, foilim=[0, 150], keeptrials=False)
tfa("welch_pow2_nooverlap", method="welch", taper="hann", t_ftimwin=0.256, toi=0.0, polyremoval=1)
save("welch_variants", **kw)

# ---------------------------------------------------------------- superlets (specest/superlet.py, compRoutines.py:655-805,
# freqanalysis.py:910-965): multiplicative and fractional adaptive, every output the path converts to
kw = {"data": np.stack(trials_of(tf)), "samplerate": tf.samplerate, "trialdefinition": tf.trialdefinition}
tfa("slt_mult", method="superlet", order_max=3, foi=np.arange(20, 90, 10), toi="all")
tfa("slt_mult_c5_toi", method="superlet", order_max=4, order_min=2, c_1=5, foi=np.array([30.0, 60.0]),
    toi=np.arange(-0.5, 0.5, 0.05), output="abs", polyremoval=1)
tfa("slt_adaptive", method="superlet", order_max=6, order_min=1, c_1=3, adaptive=True, foilim=[10, 60], toi="all",
    keeptrials=False)
tfa("slt_adaptive_fourier", method="superlet", order_max=4, adaptive=True, foi=np.arange(15, 75, 5), toi="all",
    output="fourier")
save("superlet_variants", **kw)

# ---------------------------------------------------------------- spy.mean (statistics/summary_stats.py:24-318)
kw = {}
md = synthdata.ar2_network(nTrials=6, nSamples=300, AdjMat=np.zeros((5, 5)), seed=7)
md = md + 0                                   # in-memory arithmetic copy: a plain AnalogData we may modify
md_tr = trials_of(md)
kw["data"] = np.stack(md_tr)
kw["samplerate"] = np.array(md.samplerate)
kw["trialdefinition"] = np.array(md.trialdefinition)
for name, opts in (("analog_trials", dict(dim="trials")), ("analog_time", dict(dim="time")),
                   ("analog_time_avg", dict(dim="time", keeptrials=False)), ("analog_channel", dict(dim="channel")),
                   ("analog_trials_sel", dict(dim="trials", select={"trials": [0, 2, 3], "channel": [0, 3]}))):
    r = spy.mean(md, **opts)
    kw[name] = np.array(r.data)
    kw[name + "_trldef"] = np.array(r.trialdefinition)
msp = spy.freqanalysis(md, method="mtmfft", tapsmofrq=10, keeptapers=True, output="fourier")
kw["spec"] = np.array(msp.data)
kw["spec_trldef"] = np.array(msp.trialdefinition)
for name, opts in (("spec_trials", dict(dim="trials")), ("spec_freq", dict(dim="freq")), ("spec_taper", dict(dim="taper")),
                   ("spec_channel_avg", dict(dim="channel", keeptrials=False))):
    r = spy.mean(msp, **opts)
    kw[name] = np.array(r.data)
mpw = spy.freqanalysis(md, method="mtmfft", tapsmofrq=10, output="pow")
kw["pow"] = np.array(mpw.data)
r = spy.mean(mpw, dim="trials")               # (keep the object alive while reading its HDF5 file)
kw["pow_trials"] = np.array(r.data)
r = spy.mean(mpw, dim="freq", keeptrials=False)
kw["pow_freq_avg"] = np.array(r.data)
save("mean_variants", **kw)

# ---------------------------------------------------------------- backend-level vectors
kw = {}
# harmonic known-answer signal (tests/backend/test_timefreq.py:351-404)
fs = 1000
t = np.arange(1000) / fs
sig = (5 * np.cos(2 * np.pi * 40 * t) + 3 * np.cos(2 * np.pi * 100 * t)).astype("f4")[:, None] * np.ones((1, 2), "f4")
kw["harm_sig"] = sig
ftr, fr = ref_mtmfft(sig, fs, taper=None)
kw["harm_boxcar"] = ftr
ftr, fr = ref_mtmfft(sig, fs, nSamples=1500, taper="dpss", taper_opt={"Kmax": 5, "NW": 3})
kw["harm_dpss_pad1500"] = ftr
ftr, fr = ref_mtmconvol(sig, fs, nperseg=200, noverlap=150, taper="hann")
kw["harm_stft"] = ftr
kw["harm_cwt"] = ref_wavelet(sig, fs, np.array([0.05, 0.02, 0.008]), Morlet(w0=6))
# single-trial csd incl. its norm=True branch (tests/backend/test_conn.py:88-158)
x5 = np.array(n5.trials[0])
kw["st_csd"], _ = ref_csd(x5, 200, taper="dpss", taper_opt={"Kmax": 5, "NW": 3}, norm=False)
kw["st_csd_norm"], _ = ref_csd(x5, 200, taper="dpss", taper_opt={"Kmax": 5, "NW": 3}, norm=True)
# Wilson / Granger on the trial-averaged CSD of the 5-channel network
CSD = ca(n5, method="csd", tapsmofrq=3)[0].astype(np.complex128)
H, Sigma, conv, err = wilson_sf(CSD, nIter=100, rtol=5e-6)
kw.update(w_csd=CSD, w_H=H, w_Sigma=Sigma, w_conv=np.array(conv), w_err=np.array(err))
kw["w_granger"] = granger(CSD, H, Sigma)
# regularisation of an ill-conditioned CSD (tests/backend/test_conn.py:215-242)
bad = CSD.copy()
bad[:, 4, :] = bad[:, 3, :] * (1 + 1e-7)
bad[:, :, 4] = bad[:, :, 3] * (1 + 1e-7)
reg, eps, cn0 = regularize_csd(bad, cond_max=1e4, eps_max=1e-1)
kw.update(r_in=bad, r_out=reg, r_eps=np.array(eps), r_cn0=np.array(cn0))
save("backend", **kw)

print("numpy", np.__version__, "scipy", __import__("scipy").__version__, "python", sys.version.split()[0])
