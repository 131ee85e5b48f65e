"""Satellite state holder (mirrors reference satellite.py:4-46)."""
import uuid

import numpy as np


class Satellite:
    def __init__(self, position=np.array([0., 0., 0.]), velocity=np.array([0., 0., 0.]), mass=0.):
        self.position = position    # m, ECI
        self.velocity = velocity    # m/s, ECI
        self.mass = mass            # kg
        self.id = uuid.uuid4().int

    def get_state_vector(self):
        return np.concatenate([self.position, self.velocity, np.array([self.mass])])

    def update_state_vector(self, state):
        self.position, self.velocity, self.mass = state[0:3], state[3:6], state[6]

    def __str__(self):
        return (f"Satellite {hex(self.id)} with mass {self.mass}:\n"
                f"position: {self.position}\nvelocity: {self.velocity}")
