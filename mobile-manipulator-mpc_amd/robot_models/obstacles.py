class Obstacles:
    """Ground circle obstacle record - same fields as robot_models/obstacles.py:6-10."""

    def __init__(self, x, y, radius):
        self.x = x
        self.y = y
        self.radius = radius
