"""Emulation (numpy, CPU) of the policy MLP under the operand forms against float64: plain fp32, bf16 x 3 (six products),
fp16 x 2 with the residual scaled by 2^11 into a second accumulator (round 1's form), and fp16 x 2 in scaled domains with ONE
accumulator (the shipped form: observations x 16, W1 x 16, hidden x 256, W2 x 64, outputs x 16384 -- DESIGN.md section 5) --
with and without fp16 denormal flushing, and the unscaled single-accumulator form for comparison.  Developer tool."""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf3(x):
    x = x.astype(np.float32); p0 = bf16(x); r = x - p0; p1 = bf16(r); s = r - p1; p2 = bf16(s); return p0, p1, p2
def split_h2(x, flush=False):
    x = x.astype(np.float32); h = x.astype(np.float16)
    if flush: h = np.where(np.abs(h.astype(np.float32)) < 6.1035e-5, np.float16(0), h)
    r = x - h.astype(np.float32); l = (r * np.float32(2048)).astype(np.float16)
    if flush: l = np.where(np.abs(l.astype(np.float32)) < 6.1035e-5, np.float16(0), l)
    return h.astype(np.float32), l.astype(np.float32)
def mm32(a, b):  # one MFMA-ish: exact products, fp32 accumulate (sequential over k in chunks of 32)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k0 in range(0, a.shape[1], 32):
        acc = (acc.astype(np.float64) + a[:, k0:k0+32].astype(np.float64) @ b[k0:k0+32].astype(np.float64)).astype(np.float32)
    return acc
def orth(o, i, gain):
    a = rng.standard_normal((o, i)); q, r = np.linalg.qr(a.T if o < i else a); q = q * np.sign(np.diag(r)); q = q.T if o < i else q
    return (gain * q).astype(np.float32)
D, H, A, N = 23, 256, 9, 4096
W1 = orth(H, D, np.sqrt(2)); b1 = (rng.standard_normal(H) * 0.1).astype(np.float32)
W2 = orth(A, H, 0.01 * 100).astype(np.float32); b2 = (rng.standard_normal(A) * 0.1).astype(np.float32)   # trained-like: bigger than init
X = rng.uniform(-1, 1, (N, D)).astype(np.float32); X[:, 2:4] *= rng.choice([1, 1e-3, 1e-6], (N, 1))  # tiny velocities
def truth():
    h = np.maximum(X.astype(np.float64) @ W1.T.astype(np.float64) + b1, 0); return h @ W2.T.astype(np.float64) + b2
def plain32():
    h = np.maximum(mm32(X, W1.T) + b1, 0).astype(np.float32); return mm32(h, W2.T) + b2
def bf3():
    def mm(a, b):
        A3, B3 = split_bf3(a), split_bf3(b); acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
        for i, j in [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]: acc = acc + mm32(A3[i], B3[j])
        return acc
    h = np.maximum(mm(X, W1.T) + b1, 0).astype(np.float32); return mm(h, W2.T) + b2
def h2(flush=False):
    def mm(a, b):
        ah, al = split_h2(a, flush); bh, bl = split_h2(b, flush)
        hi = mm32(ah, bh); lo = mm32(ah, bl) + mm32(al, bh)
        return (hi + lo * np.float32(2.0**-11)).astype(np.float32)
    h = np.maximum(mm(X, W1.T) + b1, 0).astype(np.float32); return mm(h, W2.T) + b2
def split_h2u(x, flush=False):   # unscaled residual
    x = x.astype(np.float32); h = x.astype(np.float16)
    if flush: h = np.where(np.abs(h.astype(np.float32)) < 6.1035e-5, np.float16(0), h)
    l = (x - h.astype(np.float32)).astype(np.float16)
    if flush: l = np.where(np.abs(l.astype(np.float32)) < 6.1035e-5, np.float16(0), l)
    return h.astype(np.float32), l.astype(np.float32)
def h2_domains(sx=16.0, s1=16.0, s2=64.0, flush=False):
    def mm(a, b, c0):   # one accumulator chain: small terms first, starting from the (scaled) bias
        ah, al = split_h2u(a, flush); bh, bl = split_h2u(b, flush)
        acc = np.broadcast_to(c0, (a.shape[0], b.shape[1])).astype(np.float32)
        for u, v in ((ah, bl), (al, bh), (ah, bh)):
            for k0 in range(0, a.shape[1], 32):
                acc = (acc.astype(np.float64) + u[:, k0:k0+32].astype(np.float64) @ v[k0:k0+32].astype(np.float64)).astype(np.float32)
        return acc
    clamp = lambda v: np.clip(v, -65504.0, 65504.0).astype(np.float32)
    xs, w1s, w2s = clamp(X * np.float32(sx)), clamp(W1 * np.float32(s1)), clamp(W2 * np.float32(s2))
    h = np.clip(mm(xs, w1s.T, b1 * np.float32(sx * s1)), 0, 65504.0).astype(np.float32)
    o = mm(h, w2s.T, np.float32(0))
    return (o * np.float32(1.0 / (sx * s1 * s2)) + b2).astype(np.float32)
t = truth()
for name, f in [("plain fp32", plain32), ("bf16x3", bf3), ("fp16x2, residual x 2^11, 2 accumulators", h2),
                ("  the same, denormals flushed", lambda: h2(True)),
                ("fp16x2, scaled domains, 1 accumulator", h2_domains), ("  the same, denormals flushed", lambda: h2_domains(flush=True)),
                ("fp16x2, unscaled, 1 accumulator", lambda: h2_domains(1.0, 1.0, 1.0))]:
    o = f(); print(f"{name:44s} max abs err {np.abs(o - t).max():.3e}  rms {np.sqrt(((o-t)**2).mean()):.3e}   max|logit| {np.abs(t).max():.2f}")
