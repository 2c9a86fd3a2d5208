"""Builds tests/golden/reference_render_256.npz from the render the reference publishes of its own sample scene.

docs/5.jpg (reference repository) is a 1024x1024 image written by the reference program itself from
scenes/sample/sample.scene - the only output of the reference that exists.  It is reduced here to 256x256 by
averaging 4x4 blocks (which also removes most of the JPEG block noise) and stored as uint8, rows top to bottom as in
the file.  docs/4.jpg (an earlier revision: different glass caustic) is stored beside it for the record.

Run in the build container, where /root/reference exists:   python tests/golden/make_docs_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
DOCS = "/root/reference/docs"


def reduce(path, n=256):
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64)
    assert a.shape == (1024, 1024, 3), a.shape
    f = a.shape[0] // n
    return np.rint(a.reshape(n, f, n, f, 3).mean(axis=(1, 3))).astype(np.uint8)


if __name__ == "__main__":
    out = os.path.join(HERE, "reference_render_256.npz")
    np.savez_compressed(out, docs5=reduce(os.path.join(DOCS, "5.jpg")), docs4=reduce(os.path.join(DOCS, "4.jpg")))
    print("wrote", out, os.path.getsize(out), "bytes")
