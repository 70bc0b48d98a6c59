from .optimizer import FusedAdam, NoamOpt
from .trainer11 import Trainer11
