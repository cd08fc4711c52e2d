"""Physical constants (mirrors reference constants.py:2)."""
c = 299_792_458.0  # speed of light in vacuum [m/s]
