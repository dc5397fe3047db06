"""Mirror of the reference's `Grid` (src/grid.rs:2-27): flat u8 buffer + width, residual of
pixel (column, line) at buffer[line * width + column].  `buffer` is a numpy array (host) or a
torch CUDA tensor (device-resident)."""


class Grid:
    def __init__(self, buffer, width):
        self.buffer = buffer.reshape(-1)
        self.width = int(width)

    @property
    def height(self):
        return self.buffer.shape[0] // self.width if self.width else 0

    def get(self, column, line):          # src/grid.rs:25-27
        return int(self.buffer[line * self.width + column])

    def set(self, at, value):             # src/grid.rs:20-22
        column, line = at
        self.buffer[line * self.width + column] = value

    def as_image(self):
        """(height, width) view of the residual plane."""
        return self.buffer.reshape(self.height, self.width)

    def __eq__(self, other):              # `#[derive(PartialEq, Eq)]`
        if not isinstance(other, Grid) or self.width != other.width:
            return False
        a, b = self.buffer, other.buffer
        if a.shape != b.shape:
            return False
        eq = a == b
        return bool(eq.all())
