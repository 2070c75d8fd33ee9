"""Hero-wavelength constants (mirror of the reference's ``spectrum/HeroSample.py:5-8``); the sampling functions themselves
(``sample``, ``sample_xyz``, ``get_rnd_hero``, ``srgb_to_spec``, ``sky_sample``) run on the device (csrc/tirt_spectral.h)."""
SAMPLE_WAVELENGTHS = 4
LAMBDA_MIN = 360.0
LAMBDA_MAX = 760.0
LAMBDA_STEP = (LAMBDA_MAX - LAMBDA_MIN) / SAMPLE_WAVELENGTHS
