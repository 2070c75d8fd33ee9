"""Regenerates the golden fixtures from the reference's committed artefacts (run in the
authoring container, where /root/reference exists; the fixtures -- data, not source --
are what travels).

  nodelist.txt            verbatim copy of /root/reference/nodelist.txt, the LBVH dump the
                          reference wrote for model/cornell_box.obj (accel/LBvh.py:164-172)
  out_png_blocks.npy      [32,32,3] f32 means of 16x16 pixel blocks of /root/reference/out.png
                          (Cornell, PT_RGB, 512^2, 512 spp, exposure 0.5; example/Example.py:49),
                          sRGB in [0,1], PNG row/col order
  veach_bdpt512_blocks.npy, veach_pt512_blocks.npy
                          the same block means of /root/reference/image/veach-bdpt512.png and veach-pt512.png, the
                          gallery renders of example/veach_bdpt.py (BDPT_RGB) and of the same scene through PT_RGB
                          (512^2; sample count and exposure not recorded by the reference: the tests fit one exposure)
"""
import shutil
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
shutil.copyfile(REF + "/nodelist.txt", "nodelist.txt")
img = np.asarray(Image.open(REF + "/out.png").convert("RGB")).astype(np.float32) / 255.0
blocks = img.reshape(32, 16, 32, 16, 3).mean(axis=(1, 3)).astype(np.float32)
np.save("out_png_blocks.npy", blocks)
print(blocks.shape, blocks.reshape(-1, 3).mean(0))
for name in ("veach-bdpt512", "veach-pt512"):
    img = np.asarray(Image.open(REF + "/image/" + name + ".png").convert("RGB")).astype(np.float32) / 255.0
    blocks = img.reshape(32, 16, 32, 16, 3).mean(axis=(1, 3)).astype(np.float32)
    np.save(name.replace("-", "_") + "_blocks.npy", blocks)
    print(name, blocks.shape, blocks.reshape(-1, 3).mean(0))
# Gallery renders of example/single_model.py's sphere (image/glass.png, metal.png, non-metal.png; 512^2).  They predate the
# committed example (measured on the images: sphere diameter 253 px = camera at 1.0 x |diagonal|, not 0.8; a light disc of
# radius 37 px at 30 deg elevation = a sphere light near (0, 2, 0) r 0.3, not (0, 20, 0) r 5; background = the env seen
# from yaw pi; a gentler tone curve than ACES(0.5 x)), so the tests that use these block means are STRUCTURE pins
# (correlation of log block luminance), not radiometric ones.
for name in ("glass", "metal", "non-metal"):
    img = np.asarray(Image.open(REF + "/image/" + name + ".png").convert("RGB")).astype(np.float32) / 255.0
    blocks = img.reshape(32, 16, 32, 16, 3).mean(axis=(1, 3)).astype(np.float32)
    np.save("gallery_" + name.replace("-", "_") + "_blocks.npy", blocks)
    print(name, blocks.shape, blocks.reshape(-1, 3).mean(0))
# image/spectral-cornellbox.png: the reference's gallery render of example/spectral_box.py (PT_Spec); tests/test_spectral.py says why
# it can only be a structure pin
img = np.asarray(Image.open(REF + "/image/spectral-cornellbox.png").convert("RGB")).astype(np.float32) / 255.0
np.save("spectral_cornellbox_blocks.npy", img.reshape(32, 16, 32, 16, 3).mean(axis=(1, 3)).astype(np.float32))

# image/rainbow.png: the gallery render of example/prism_rainbow.py (BDPT_SPEC: a laser through a glass prism, the spectrum on the far wall).
# Taken with the close-up camera that the example keeps as comments (yaw 0.8, scale 20, target (-50, 2, -93): prism_rainbow.py:58-60), and
# black outside the spectrum: tests/test_bdpt_spec.py renders the laser alone.  64 x 64 block means (8 x 8 pixels), sRGB in [0,1].
img = np.asarray(Image.open(REF + "/image/rainbow.png").convert("RGB")).astype(np.float32) / 255.0
np.save("rainbow_blocks.npy", img.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3)).astype(np.float32))
