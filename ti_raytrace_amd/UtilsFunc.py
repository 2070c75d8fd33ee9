"""Host-visible part of the reference's ``UtilsFunc.py``.

In the reference this module is a bag of ``@ti.func`` device helpers plus one kernel,
``tone_map``.  The device helpers live in ``csrc/tirt_device.h`` here; what a host script
can call is ``tone_map`` and the constants.
"""
AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2
EPS = 0.00001
M_PIf = 3.1415956            # sic, UtilsFunc.py:37 (quirk B1)
INF_VALUE = 1000000.0


def tone_map(exposure, input, output):
    """``output = lrgb_to_srgb(tone_ACES(input * exposure))`` (UtilsFunc.py:583-586).

    ``input`` / ``output`` are the ``hdr`` / ``rgb_film`` fields of one integrator."""
    if input.ctx is not output.ctx or input.name != "hdr" or output.name != "rgb_film":
        raise ValueError("tone_map expects (exposure, integrator.hdr, integrator.rgb_film)")
    input.ctx.tone_map(exposure)
