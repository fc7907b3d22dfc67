"""Unit system of the multi-rods model (reference: constants.py:5-12).

Energies are measured in units where hbar^2/2m = 1 and the lattice period is
the unit of length, so the recoil energy of the lattice is pi^2.
"""
from math import pi

UE = 1.0
ER = pi ** 2 * UE
ER_UE = ER / UE
LKP = 1.0
K_OPT = pi / LKP
