"""numpy restatement of cv2.bgsegm.createBackgroundSubtractorMOG(...).apply on 8-bit 3-channel images.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED: cv2 (opencv-contrib) is absent here, the reference holds
no model state and no mask of this stage alone, and its training video cannot be decoded here; this restates the published
algorithm of opencv_contrib's bgsegm module (bgfg_gaussmix.cpp: BackgroundSubtractorMOGImpl, process8uC3 -- KaewTraKulPong &
Bowden's mixture of Gaussians).  Reference lines it stands for: background_subtraction.py:75-92 (the model and its training
loop, `apply(frame, None, learning_rate)` with learning_rate -1) and :158 (`bg_model.apply(image, None, 0)`).

Constructor (non-positive arguments select the defaults): nmixtures = min(n > 0 ? n : 5, 8); history = h > 0 ? h : 200;
varThreshold = 2.5 * 2.5; backgroundRatio = min(r > 0 ? r : 0.95, 1); noiseSigma = s <= 0 ? 15 : s.
apply(image, learningRate): the model (K x {sortKey, weight, mean[3], var[3]} float32 per pixel, all zero) is created on the
first frame, on learningRate >= 1 and on a change of image size; ++nframes; alpha = learningRate if learningRate >= 0 and
nframes > 1 else 1 / min(nframes, history).  Per pixel, float32, operations in this order (no contraction):

  alpha > 0:  wsum = 0; for k in 0..K-1: w = weight[k]; wsum += w; if w < FLT_EPSILON: break
                  diff = pix - mean[k]; d2 = (diff0^2 + diff1^2) + diff2^2
                  if d2 < vT * ((var0 + var1) + var2):                                   # the component takes the pixel
                      wsum -= w; weight[k] = w + alpha * (1 - w); mean[k] += alpha * diff
                      var[k] = max(var + alpha * (diff^2 - var), noiseSigma^2); sortKey[k] = w / sqrt((var0 + var1) + var2)
                      bubble k up while sortKey[k1] < sortKey[k1 + 1]; kHit = where it lands; break
              no hit: k = min(k, K-1); wsum += w0 - weight[k]; component k = {w0 = 0.05, pix, var0 = (2 * 15)^2, sk0 = w0 / (30 sqrt 3)}
              hit:    wsum += weight[k'] for k' = k .. K-1          (positions AFTER the bubbling, as the code does)
              all weights and sort keys *= 1 / wsum; kForeground = 1 + the first k whose running weight sum exceeds backgroundRatio
              mask = 255 if kHit >= kForeground else 0
  alpha == 0: nothing is written: kHit = first component (weight >= FLT_EPSILON) within the threshold; mask = 255 if there is none
              or kHit >= kForeground (running weight sum as above; -1 if it never exceeds the ratio)
"""
import numpy as np

F = np.float32
EPS = F(1.1920928955078125e-7)


class MOG:
    def __init__(self, history=200, nmixtures=5, backgroundRatio=0.7, noiseSigma=0):
        self.K = min(nmixtures if nmixtures > 0 else 5, 8)
        self.history = history if history > 0 else 200
        self.T = F(min(backgroundRatio if backgroundRatio > 0 else 0.95, 1.0))
        self.noise_sigma = 15.0 if noiseSigma <= 0 else float(noiseSigma)
        self.vT = F(2.5 * 2.5)
        self.nframes = 0
        self.shape = None
        self.state = None          # [8 K][H W]: plane 8 k + f (f: 0 sortKey, 1 weight, 2..4 mean, 5..7 var)

    def apply(self, image, learningRate=-1):
        img = np.asarray(image, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 3
        if self.nframes == 0 or learningRate >= 1 or img.shape[:2] != self.shape:
            self.shape = img.shape[:2]
            self.state = np.zeros((8 * self.K, img.shape[0] * img.shape[1]), F)
            self.nframes = 0
        self.nframes += 1
        lr = learningRate if learningRate >= 0 and self.nframes > 1 else 1.0 / min(self.nframes, self.history)
        return self._process(img.reshape(-1, 3).astype(F), F(lr)).reshape(self.shape)

    def _process(self, pix, alpha):
        K, T, vT = self.K, self.T, self.vT
        st = self.state.reshape(K, 8, -1)
        sk, w, mu, var = st[:, 0], st[:, 1], st[:, 2:5], st[:, 5:8]          # views: [K, n], [K, n], [K, 3, n], [K, 3, n]
        n = pix.shape[0]
        pix = pix.T                                                           # [3, n]
        w0 = F(0.05)
        sk0 = F(0.05 / (15.0 * 2 * np.sqrt(3.0)))
        var0 = F(15.0 * 15.0 * 4)
        min_var = F(self.noise_sigma * self.noise_sigma)
        k_hit = np.full(n, -1, np.int64)
        k_fg = np.full(n, -1, np.int64)
        if alpha > 0:
            wsum = np.zeros(n, F)
            kstop = np.full(n, K, np.int64)
            done = np.zeros(n, bool)
            for k in range(K):
                act = ~done
                wk = w[k].copy()
                wsum = np.where(act, wsum + wk, wsum)
                empty = act & (wk < EPS)
                kstop[empty] = k
                done |= empty
                act &= ~empty
                d = pix - mu[k]
                d2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
                hit = act & (d2 < vT * ((var[k, 0] + var[k, 1]) + var[k, 2]))
                if hit.any():
                    wsum = np.where(hit, wsum - wk, wsum)
                    w[k] = np.where(hit, wk + alpha * (F(1) - wk), w[k])
                    for c in range(3):
                        mu[k, c] = np.where(hit, mu[k, c] + alpha * d[c], mu[k, c])
                        nv = var[k, c] + alpha * (d[c] * d[c] - var[k, c])
                        var[k, c] = np.where(hit, np.maximum(nv, min_var), var[k, c])
                    with np.errstate(divide="ignore", invalid="ignore"):
                        sk[k] = np.where(hit, wk / np.sqrt((var[k, 0] + var[k, 1]) + var[k, 2]), sk[k])
                    pos = np.full(n, k, np.int64)
                    moving = hit.copy()
                    for k1 in range(k - 1, -1, -1):
                        swap = moving & ~(sk[k1] >= sk[k1 + 1])
                        moving = swap
                        for arr in (sk, w):
                            a, b = arr[k1].copy(), arr[k1 + 1].copy()
                            arr[k1] = np.where(swap, b, a); arr[k1 + 1] = np.where(swap, a, b)
                        for arr in (mu, var):
                            a, b = arr[k1].copy(), arr[k1 + 1].copy()
                            arr[k1] = np.where(swap, b, a); arr[k1 + 1] = np.where(swap, a, b)
                        pos[swap] = k1
                    k_hit[hit] = pos[hit]
                    kstop[hit] = k
                    done |= hit
            miss = k_hit < 0
            kk = np.minimum(kstop, K - 1)
            for k in range(K):
                m = miss & (kk == k)
                if m.any():
                    wsum = np.where(m, wsum + (w0 - w[k]), wsum)
                    w[k] = np.where(m, w0, w[k]); sk[k] = np.where(m, sk0, sk[k])
                    for c in range(3):
                        mu[k, c] = np.where(m, pix[c], mu[k, c]); var[k, c] = np.where(m, var0, var[k, c])
                wsum = np.where(~miss & (k >= kstop), wsum + w[k], wsum)
            k_hit[miss] = kk[miss]
            with np.errstate(divide="ignore"):
                wscale = F(1) / wsum
            wsum = np.zeros(n, F)
            for k in range(K):
                w[k] = w[k] * wscale
                wsum = wsum + w[k]
                sk[k] = sk[k] * wscale
                k_fg = np.where((wsum > T) & (k_fg < 0), k + 1, k_fg)
            return np.where(k_hit >= k_fg, 255, 0).astype(np.uint8)
        done = np.zeros(n, bool)
        for k in range(K):
            act = ~done
            empty = act & (w[k] < EPS)
            done |= empty
            act &= ~empty
            d = pix - mu[k]
            d2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
            hit = act & (d2 < vT * ((var[k, 0] + var[k, 1]) + var[k, 2]))
            k_hit[hit] = k
            done |= hit
        wsum = np.zeros(n, F)
        for k in range(K):
            open_ = k_fg < 0
            wsum = np.where(open_, wsum + w[k], wsum)
            k_fg = np.where(open_ & (wsum > T), k + 1, k_fg)
        k_fg = np.where(k_hit >= 0, k_fg, -1)
        return np.where((k_hit < 0) | (k_hit >= k_fg), 255, 0).astype(np.uint8)


class MOGLiteral(MOG):
    """The same model pixel by pixel, line for line as the description above (small images: cross-check of the vectorised form)."""

    def _process(self, pix, alpha):
        K, T, vT = self.K, self.T, self.vT
        st = self.state.reshape(K, 8, -1)
        w0 = F(0.05)
        sk0 = F(0.05 / (15.0 * 2 * np.sqrt(3.0)))
        var0 = F(15.0 * 15.0 * 4)
        min_var = F(self.noise_sigma * self.noise_sigma)
        out = np.zeros(pix.shape[0], np.uint8)
        for i in range(pix.shape[0]):
            m = st[:, :, i].copy()                                           # [K, 8]
            p = pix[i]
            k_hit = k_fg = -1
            if alpha > 0:
                wsum = F(0)
                k = 0
                while k < K:
                    wk = m[k, 1]
                    wsum = F(wsum + wk)
                    if wk < EPS:
                        break
                    d = (p - m[k, 2:5]).astype(F)
                    d2 = F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
                    if d2 < F(vT * F(F(m[k, 5] + m[k, 6]) + m[k, 7])):
                        wsum = F(wsum - wk)
                        m[k, 1] = F(wk + F(alpha * F(F(1) - wk)))
                        for c in range(3):
                            m[k, 2 + c] = F(m[k, 2 + c] + F(alpha * d[c]))
                            m[k, 5 + c] = max(F(m[k, 5 + c] + F(alpha * F(F(d[c] * d[c]) - m[k, 5 + c]))), min_var)
                        m[k, 0] = F(wk / np.sqrt(F(F(m[k, 5] + m[k, 6]) + m[k, 7])))
                        k1 = k - 1
                        while k1 >= 0:
                            if m[k1, 0] >= m[k1 + 1, 0]:
                                break
                            m[[k1, k1 + 1]] = m[[k1 + 1, k1]]
                            k1 -= 1
                        k_hit = k1 + 1
                        break
                    k += 1
                if k_hit < 0:
                    k_hit = k = min(k, K - 1)
                    wsum = F(wsum + F(w0 - m[k, 1]))
                    m[k] = [sk0, w0, p[0], p[1], p[2], var0, var0, var0]
                else:
                    while k < K:
                        wsum = F(wsum + m[k, 1])
                        k += 1
                with np.errstate(divide="ignore"):
                    wscale = F(F(1) / wsum)
                wsum = F(0)
                for k in range(K):
                    m[k, 1] = F(m[k, 1] * wscale)
                    wsum = F(wsum + m[k, 1])
                    m[k, 0] = F(m[k, 0] * wscale)
                    if wsum > T and k_fg < 0:
                        k_fg = k + 1
                out[i] = 255 if k_hit >= k_fg else 0
                st[:, :, i] = m
            else:
                for k in range(K):
                    if m[k, 1] < EPS:
                        break
                    d = (p - m[k, 2:5]).astype(F)
                    d2 = F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
                    if d2 < F(vT * F(F(m[k, 5] + m[k, 6]) + m[k, 7])):
                        k_hit = k
                        break
                if k_hit >= 0:
                    wsum = F(0)
                    for k in range(K):
                        wsum = F(wsum + m[k, 1])
                        if wsum > T:
                            k_fg = k + 1
                            break
                out[i] = 255 if (k_hit < 0 or k_hit >= k_fg) else 0
        return out
