"""The grid-position value type the N=1 classes hand out as `agent_pos` and list in `ACTIONS`.

Counterpart of the reference's `Coord` (gym_craftingworld/envs/coordinates.py:6-42; `agent_pos` is built at
craftingworld_ray.py:624-626 with max_row = STATE_W - 1, max_col = STATE_H - 1): attributes row / col / max_row /
max_col / name, `+` and `-` clamped to [0, max_row] x [0, max_col] (coordinates.py:22-30), `tuple()`, `str()` of
the tuple.  On the device the same clamp is two min/max in step_env (csrc/cw_kernels.hip); this class is host
convenience for code written against the reference's attribute.

Two deliberate extensions: it compares equal to a plain (row, col) pair as well as to another position (the
reference's __eq__ is False for anything but a Coord; rounds 1-4 of this package returned a tuple here), and it
unpacks / indexes like that pair (`r, c = env.agent_pos`).
"""


class GridPos:
    __slots__ = ('row', 'col', 'max_row', 'max_col', 'name')

    def __init__(self, row, col, max_row=100, max_col=100, name=None):
        self.row, self.col = int(row), int(col)
        self.max_row, self.max_col = int(max_row), int(max_col)
        self.name = name

    def _moved(self, d_row, d_col):
        return GridPos(min(max(self.row + d_row, 0), self.max_row), min(max(self.col + d_col, 0), self.max_col),
                       self.max_row, self.max_col)

    def __add__(self, other):
        return self._moved(other.row, other.col)

    def __sub__(self, other):
        return self._moved(-other.row, -other.col)

    def tuple(self):
        return (self.row, self.col)

    # -- the (row, col) pair it stands for
    def __iter__(self):
        return iter((self.row, self.col))

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return (self.row, self.col)[i]

    def __eq__(self, other):
        if hasattr(other, 'row') and hasattr(other, 'col'):
            return self.row == other.row and self.col == other.col
        try:
            return len(other) == 2 and self.row == other[0] and self.col == other[1]
        except TypeError:
            return False

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self.row, self.col))

    def __str__(self):
        return str(self.tuple())

    def __repr__(self):
        return 'GridPos(%d, %d)' % (self.row, self.col) if self.name is None else 'GridPos(%d, %d, name=%r)' % (self.row, self.col, self.name)


Coord = GridPos          # the reference's name for it (coordinates.py:6)
