"""Host (numpy) statement of the synthetic speech-like signal generated on the device by
vbx_synth_speech_f64 (csrc/k_synth.hip).  Used by the CPU tests and the bench's CPU-baseline
sample; the two generators agree to rounding (they are compared in tests/test_gpu_synth.py),
but parity tests always pull the DEVICE samples back so both sides see identical input."""
import numpy as np

F_FORMANT = np.array([700.0, 1220.0, 2600.0, 3300.0])
HALF_BW = np.array([65.0, 35.0, 80.0, 125.0])
SEED = 0x5EED0001
_M64 = (1 << 64) - 1


def _splitmix_uniform(s, seed):
    z = (np.uint64(seed) + s.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return 2.0 * ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)) - 1.0


def synth_speech(n_samples, sample_offset=0, sample_rate=48000.0, seed=SEED):
    with np.errstate(over="ignore"):
        s = np.arange(sample_offset, sample_offset + n_samples, dtype=np.uint64)
        noise = _splitmix_uniform(s, seed)
    t = s.astype(np.float64) / sample_rate
    isec = np.floor(t).astype(np.int64)
    unvoiced = (isec % 5) == 4
    tau = t - 2.0 * np.floor(t * 0.5)
    up = tau < 1.0
    u = tau - 1.0
    f0 = np.where(up, 90.0 + 160.0 * tau, 250.0 - 160.0 * u)
    phi = np.where(up, 90.0 * tau + 80.0 * tau * tau, 170.0 + 250.0 * u - 80.0 * u * u)
    th = 2.0 * np.pi * (phi - np.floor(phi))
    sn, cs = np.sin(th), np.cos(th)
    two_c = 2.0 * cs
    s_prev, s_cur = np.zeros_like(th), sn
    acc = np.zeros_like(th)
    for h in range(1, 31):
        fh = h * f0
        g = np.zeros_like(th)
        for k in range(4):
            d = (fh - F_FORMANT[k]) / HALF_BW[k]
            g += 1.0 / np.sqrt(1.0 + d * d)
        acc += (g / h) * s_cur
        s_prev, s_cur = s_cur, two_c * s_cur - s_prev
    return np.where(unvoiced, 0.05 * noise, 0.25 * acc + 0.0025 * noise)



def speech_recording(torch, device, pcm16, n_samples, seed=0x5EED0044, dither_db=-70.0):
    """A long recording of REAL speech for benches and soaks, built on `device` with torch (plumbing): the 16-bit samples of
    a WAV fixture (what a reader hands the reference's callers, tests/lib.rs:15-19) tiled to n_samples -- every tile with its
    own gain in [0.5, 1) (the golden ratio's multiples mod 1) and the whole under a uniform dither at dither_db re full scale,
    so that no two frames are the same bits -- as f64 in [-1, 1]: `sample / 32767` first, exactly as the reference's test
    converts.  The dither comes from torch's generator on that device (seeded): a test that needs the samples on the host
    copies them back, both sides then see identical input."""
    f64 = torch.float64
    base = torch.as_tensor(np.asarray(pcm16, dtype=np.float64) / 32767.0, dtype=f64, device=device)
    L = int(base.numel())
    tiles = -(-int(n_samples) // L)
    gains = torch.as_tensor(0.5 + 0.5 * np.modf(0.6180339887498949 * np.arange(1, tiles + 1))[0], dtype=f64, device=device)
    out = (gains[:, None] * base[None, :]).reshape(-1)[:n_samples].contiguous()
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    amp = 10.0 ** (dither_db / 20.0)
    step = 1 << 26
    for s0 in range(0, int(n_samples), step):
        m = min(step, int(n_samples) - s0)
        out[s0:s0 + m] += amp * (2.0 * torch.rand(m, generator=gen, dtype=f64, device=device) - 1.0)
    return out
