import numpy as np


def HWCtoCHW(x):
    return np.transpose(x, (2, 0, 1))


def CHWtoHWC(x):
    return np.transpose(x, (1, 2, 0))
