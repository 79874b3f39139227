from .base import Scene
from .sequential import SequentialScene
