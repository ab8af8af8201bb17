from .eig import EIGStepLoss
