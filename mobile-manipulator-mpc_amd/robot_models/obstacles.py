"""Ground circle obstacle record with the field names the controllers read (`x`, `y`, `radius`;
reference: robot_models/obstacles.py:6-10).  A dataclass, so that it also compares, prints and unpacks."""
import dataclasses


@dataclasses.dataclass
class Obstacles:
    x: float
    y: float
    radius: float

    def as_row(self):
        """(x, y, radius) - one row of the obs[B][M][3] array of include/mmpc.h"""
        return (float(self.x), float(self.y), float(self.radius))
