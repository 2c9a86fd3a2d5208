"""Builds tests/golden/reference_render_256.npz and reference_render_512.npz from the renders the reference publishes
of its own sample scene.

docs/5.jpg (reference repository) is a 1024x1024 image written by the reference program itself from
scenes/sample/sample.scene - the only output of the current reference revision that exists.  It is reduced here to
256x256 (averaging 4x4 blocks, which also removes most of the JPEG block noise) and to 512x512 (2x2 blocks) and stored
as uint8, rows top to bottom as in the file.  Stored beside it for the record and as controls a render of this scene
must be FURTHER from: docs/4.jpg (an earlier revision: different glass caustic, darker walls) and docs/3.jpg (512x512;
another earlier revision - it is 14-16 RMSE units from both 4.jpg and 5.jpg at every scale from 512 down to 64 pixels,
a systematic difference around the glossy sphere and the ceiling, not Monte Carlo noise - so it cannot pin the code
that wrote 5.jpg).

Run in the build container, where /root/reference exists:   python tests/golden/make_docs_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
DOCS = "/root/reference/docs"


def reduce(path, n=256):
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64)
    assert a.shape[0] == a.shape[1] and a.shape[0] % n == 0 and a.shape[2] == 3, a.shape
    f = a.shape[0] // n
    return np.rint(a.reshape(n, f, n, f, 3).mean(axis=(1, 3))).astype(np.uint8)


if __name__ == "__main__":
    out = os.path.join(HERE, "reference_render_256.npz")
    np.savez_compressed(out, docs5=reduce(os.path.join(DOCS, "5.jpg")), docs4=reduce(os.path.join(DOCS, "4.jpg")),
                        docs3=reduce(os.path.join(DOCS, "3.jpg")))
    print("wrote", out, os.path.getsize(out), "bytes")
    out = os.path.join(HERE, "reference_render_512.npz")
    np.savez_compressed(out, docs5=reduce(os.path.join(DOCS, "5.jpg"), 512))
    print("wrote", out, os.path.getsize(out), "bytes")
