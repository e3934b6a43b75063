"""Prototype (NumPy, CPU) of the zero-state pass of the fused time-parallel cascade as a GEMM -- DESIGN.md section 7, item 2.

k_cscan today: prep -> pass 1 (the whole cascade over every chunk, no stores: chunk end states) -> fix (S_{j+1} = M S_j + z_j)
-> pass 2.  Pass 1 issues ~28 FP64 instructions per sample only to leave 8 numbers per chunk.  Those 8 numbers are linear in the
chunk's input:

    z_j[s][k] = y_s[L - 1 - k]  (zero output history, true input history)
              = sum_{m=0}^{L-1-k} h_s[L - 1 - k - m] x_j[m]  +  e1[s][k] x_j[-1]  +  e2[s][k] x_j[-2]

with h_s the impulse response of sections 0 .. s and e1 / e2 the responses (at the same two instants) to a unit sample sitting in
section 0's input history.  For one input stream that is E[band * 8, chunk] = H[band * 8, L] . X[L, chunk] (+ a rank-2 edge term).
This script checks the formulation against a plain serial run: end states of the zero-state pass, the carried states after the
fix recursion (chunk 0 from a self-consistent bank state), and the outputs of the replay -- on gammatone-like cascades.

    python tools/experiments/cscan_gemm_prototype.py [bands] [chunks] [chunk_len]
"""
import sys
import numpy as np

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 512
NSEC = 4
rng = np.random.default_rng(7)

# four biquad sections per band: poles of a 4th-order gammatone-like band (same pole pair in every section), two zeros each
fc = np.geomspace(50., 16000., B) * 2 * np.pi / 48000.
bw = 1.019 * 24.7 * (4.37 * fc * 48000. / (2000. * np.pi) + 1) * 2 * np.pi / 48000.
r = np.exp(-bw)
a1 = np.tile(-2 * r * np.cos(fc), (NSEC, 1))
a2 = np.tile(r * r, (NSEC, 1))
b0 = rng.uniform(.5, 1., (NSEC, B)); b1 = rng.uniform(-1., 1., (NSEC, B)) * r; b2 = rng.uniform(-.5, .5, (NSEC, B))


def run(x, xh, yh, record=False):
  """The cascade on x [n] for every band.  xh [NSEC, 2, B] input history, yh [NSEC, 2, B] output history (index 0 = newest).
  Returns (y [n, B], xh, yh[, per-section outputs [n, NSEC, B]])."""
  xh, yh = xh.copy(), yh.copy()
  out = np.empty((len(x), B)); rec = np.empty((len(x), NSEC, B)) if record else None
  for n in range(len(x)):
    v = np.full(B, x[n])
    for s in range(NSEC):
      y = ((b0[s] * v + b1[s] * xh[s, 0]) + b2[s] * xh[s, 1]) - a1[s] * yh[s, 0] - a2[s] * yh[s, 1]
      xh[s, 1] = xh[s, 0]; xh[s, 0] = v
      yh[s, 1] = yh[s, 0]; yh[s, 0] = y
      v = y
      if record: rec[n, s] = y
    out[n] = v
  return (out, xh, yh, rec) if record else (out, xh, yh)


N = K * L
x = rng.uniform(-1., 1., N)
zero = np.zeros((NSEC, 2, B))
# a self-consistent bank state: what a previous block leaves behind (section s + 1's input history = section s's output history)
_, xh0, yh0 = run(rng.uniform(-1., 1., 300), zero, zero)
y_true, _, _ = run(x, xh0, yh0)

# ---- tables (once per bank and chunk length) ----
imp = np.zeros(L); imp[0] = 1.
_, _, _, h = run(imp, zero, zero, record=True)                  # h[t, s, band]
edge = np.empty((2, NSEC, 2, B))                                # edge[q][s][k]: x[-1-q] = 1 -> y_s[L-1-k]
for q in range(2):
  xh = zero.copy(); xh[0, q] = 1.
  _, _, _, rec = run(np.zeros(L), xh, zero, record=True)
  edge[q, :, 0] = rec[L - 1]; edge[q, :, 1] = rec[L - 2]
M = np.empty((2 * NSEC, 2 * NSEC, B))                           # column e: zero input, unit output state e (ties the next section's input history to it)
for e in range(2 * NSEC):
  yh = zero.copy(); yh[e // 2, e % 2] = 1.
  xh = zero.copy(); xh[1:] = yh[:-1]
  _, _, yl = run(np.zeros(L), xh, yh)
  M[:, e] = yl.reshape(2 * NSEC, B)

# ---- the zero-state pass as a GEMM ----
X = x.reshape(K, L).T                                           # [L, chunk]
H0 = h[::-1].transpose(1, 2, 0)                                 # H0[s, band, m] = h_s[L - 1 - m]
H1 = np.concatenate([h[-2::-1], np.zeros((1, NSEC, B))]).transpose(1, 2, 0)   # h_s[L - 2 - m], 0 at m = L - 1
Hm = np.stack([H0, H1], axis=1).reshape(2 * NSEC, B, L)         # rows r = 2 s + k
E = np.einsum('rbm,mj->rbj', Hm, X)                             # [8, band, chunk]  -- the GEMM
prev = np.concatenate([[xh0[0, 0, 0]], x[L - 1:N - 1:L]])       # x_j[-1]   (chunk 0: the bank's input history; one stream)
prev2 = np.concatenate([[xh0[0, 1, 0]], x[L - 2:N - 2:L]])      # x_j[-2]
E += edge[0].reshape(2 * NSEC, B)[:, :, None] * prev[None, None, :] + edge[1].reshape(2 * NSEC, B)[:, :, None] * prev2[None, None, :]

# reference zero-state end states: the cascade itself on every chunk (what pass 1 computes today)
Z = np.empty_like(E)
for j in range(K):
  xh = zero.copy(); xh[0, 0] = prev[j]; xh[0, 1] = prev2[j]
  _, _, yl = run(x[j * L:(j + 1) * L], xh, zero)
  Z[:, :, j] = yl.reshape(2 * NSEC, B)
scale = np.abs(Z).max()
print("zero-state end states, GEMM against the cascade: max |diff| / max |z| = %.2e" % (np.abs(E - Z).max() / scale))

# ---- fix: S_{j+1} = M S_j + z_j from the bank's state (chunk 0 is no special case when the state is self-consistent) ----
S = yh0.reshape(2 * NSEC, B).copy()
worst = 0.
y_tp = np.empty_like(y_true)
for j in range(K):
  yh = S.reshape(NSEC, 2, B)
  xh = zero.copy(); xh[1:] = yh[:-1]; xh[0, 0] = prev[j]; xh[0, 1] = prev2[j]
  y_tp[j * L:(j + 1) * L], _, yl = run(x[j * L:(j + 1) * L], xh, yh)         # pass 2 (replay) from the carried state
  S_next = np.einsum('reb,eb->rb', M, S) + E[:, :, j]
  worst = max(worst, np.abs(S_next - yl.reshape(2 * NSEC, B)).max() / scale)
  S = S_next
print("carried states after the fix recursion against the replay's own end states: %.2e" % worst)
print("outputs, time-parallel against serial: max |diff| / max |y| = %.2e  (contract 1e-9)" % (np.abs(y_tp - y_true).max() / np.abs(y_true).max()))
print("work per output sample: 8 multiply-adds (GEMM %d x %d x %d = %.2f GFLOP) against ~28 instructions of the cascade pass"
      % (2 * NSEC * B, L, K, 2. * 2 * NSEC * B * L * K / 1e9))
