"""
pyro_amd -- MI355X-native grid value iteration behind pyro's class surface.

Host side (this package) mirrors the reference interface of the hot path only:
  pyro_amd.dynamic.{system,mechanical,pendulum,cartpole,manipulator}
  pyro_amd.analysis.costfunction
  pyro_amd.planning.{discretizer,dynamicprogramming}
  pyro_amd.control.controller        (just enough for LookUpTableController / ctl + sys)
Device side: pyro_amd/csrc/{pyrovi,f64,lean}.hip (+ core.h, host.h) -> pyro_amd/libpyrovi.so, bound with ctypes in
pyro_amd._native (C ABI declared in include/pyrovi.h).
"""
__version__ = "0.1.0"
