"""Environment / albedo texture (mirror of the reference's ``texture/Texture.py``).

The reference loads with ``cv2.imread`` (BGR, alpha dropped) and packs each texel into one
i32 ``0xRRGGBB`` stored x-major with y flipped (``texture/Texture.py:18-34``).  cv2 is not
available here; PIL's ``convert('RGB')`` yields the same 8-bit channels.  Sampling
(``sample`` / ``texture2D``, :41-69) runs on the device inside the miss branch of
``PT_RGB.render``.
"""
import numpy as np


class Texture:
    def __init__(self):
        self.wid = 0
        self.hgt = 0
        self.channel = 0
        self.size = 0
        self.np_img = None

    def load_image(self, imagePath):
        from PIL import Image
        img = np.asarray(Image.open(imagePath).convert("RGB"), dtype=np.int32)   # [hgt, wid, 3] RGB
        self.load_array(img)

    def load_array(self, rgb_u8):
        """rgb_u8: [hgt, wid, 3] integer array, row 0 = top of the image."""
        img = np.asarray(rgb_u8).astype(np.int32)
        self.hgt, self.wid, self.channel = img.shape[0], img.shape[1], 3
        self.size = self.wid * self.hgt * self.channel
        packed = (img[:, :, 0] << 16) | (img[:, :, 1] << 8) | img[:, :, 2]          # [hgt, wid]
        # np_img[j, hgt-1-i] = packed[i, j]   (texture/Texture.py:29-34)
        self.np_img = np.ascontiguousarray(packed[::-1, :].T, dtype=np.int32)       # [wid, hgt]

    def load_black(self, wid=512, hgt=512):
        """Equivalent of the reference's default ``image/black.png`` (Scene.py:295-296)."""
        self.load_array(np.zeros((hgt, wid, 3), np.int32))

    def setup_data_gpu(self, ctx, power):
        ctx.env_upload(self.np_img, power)
