"""Seeded synthetic baseline-JPEG corpora: bicubic-upsampled 64-px random colour field
+ bilinear-upsampled N(0,18) quarter-resolution texture + N(0,4) per-pixel noise, saved with PIL at
quality 90, 4:2:0, optimize=False.  Bytes depend on the bundled libjpeg, so a corpus is generated once
per run and the same bytes go to every implementation under test."""
import io
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np


def synth_jpeg(w, h, seed, quality=90, progressive=False, subsampling="4:2:0", skew=0.0):
    from PIL import Image

    rng = np.random.default_rng(seed)
    cw, ch = max(2, (w + 63) // 64), max(2, (h + 63) // 64)
    base = Image.fromarray(rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8), "RGB").resize((w, h), Image.BICUBIC)
    img = np.asarray(base, dtype=np.float32)
    qw, qh = max(1, w // 4), max(1, h // 4)
    tex = rng.normal(0.0, 18.0, (qh, qw, 3)).astype(np.float32)
    tex_img = np.stack(
        [np.asarray(Image.fromarray(tex[:, :, c], "F").resize((w, h), Image.BILINEAR)) for c in range(3)], axis=2
    )
    detail = tex_img + rng.normal(0.0, 4.0, (h, w, 3)).astype(np.float32)
    if skew:   # photograph-like: smooth at the top ("sky"), detail growing towards the bottom -> thread segments of equal
        #        compressed size then differ several-fold in blocks, as they do for real pictures
        ramp = ((np.arange(h, dtype=np.float32) + 0.5) / h) ** float(skew)
        detail *= ramp[:, None, None]
    img = img + detail
    out = Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")
    buf = io.BytesIO()
    out.save(buf, format="JPEG", quality=quality, subsampling=subsampling, optimize=False, progressive=progressive)
    return buf.getvalue()


def _job(args):
    return synth_jpeg(*args)


def make_corpus(n, w, h, seed0, workers=None, **kw):
    """n distinct images, seeds seed0..seed0+n-1, generated with a process pool."""
    jobs = [(w, h, seed0 + i, kw.get("quality", 90), kw.get("progressive", False), kw.get("subsampling", "4:2:0"), kw.get("skew", 0.0)) for i in range(n)]
    workers = workers or min(len(jobs), max(1, (os.cpu_count() or 2) - 1), 32)
    if workers <= 1 or n <= 2:
        return [_job(j) for j in jobs]
    with ProcessPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(_job, jobs, chunksize=1))
