"""Host mirrors of pyro/dynamic/massspringdamper.py: SingleMass (:16-63) and FloatingSingleMass (:731-745),
linear state-space systems in mechanical form (position, velocity; force input).  Drawing code is out of scope."""
import numpy as np

from pyro_amd.dynamic import statespace


class SingleMass(statespace.StateSpaceSystem):
    """m ddx + b dx + k x = u."""

    def __init__(self, mass=1, k=2, b=0):
        self.mass, self.k, self.b = mass, k, b
        self.l1, self.l2 = 2, 1
        self.compute_ABCD()
        statespace.StateSpaceSystem.__init__(self, self.A, self.B, self.C, self.D)
        self.name = "Linear-Spring-Damper"
        self.input_label, self.input_units = ["Force"], ["[N]"]
        self.output_label, self.output_units = ["Position"], ["[m]"]
        self.state_label, self.state_units = ["Position", "Velocity"], ["[m]", "[m/s]"]

    def compute_ABCD(self):
        self.A = np.array([[0, 1], [-self.k / self.mass, -self.b / self.mass]])
        self.B = np.array([[0], [1 / self.mass]])
        self.C = np.array([[1, 0]])
        self.D = np.array([[0]])


class FloatingSingleMass(SingleMass):
    """A mass without spring: m ddx + b dx = u."""

    def __init__(self, m=1, b=0):
        SingleMass.__init__(self, m, 0, b)
        self.name = "Mass"
