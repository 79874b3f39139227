from .goals import Goal, FocalLengthLoss, SpotSizeLoss, SpotTargetLoss
from .constraints import Constraint, ThicknessConstraint, SpacingConstraint, SystemLengthConstraint
