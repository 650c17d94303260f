"""MultiGrid stand-in: per-cell python lists, append on place, (content,(x,y)) coord_iter."""


class MultiGrid:
    def __init__(self, width, height, torus):
        self.width = width
        self.height = height
        self.torus = torus
        self._grid = [[[] for _ in range(height)] for _ in range(width)]

    def __getitem__(self, index):
        x, y = index
        return self._grid[x][y]

    def out_of_bounds(self, pos):
        x, y = pos
        return x < 0 or x >= self.width or y < 0 or y >= self.height

    def place_agent(self, agent, pos):
        x, y = pos
        if agent.pos is None or agent not in self._grid[x][y]:
            self._grid[x][y].append(agent)
            agent.pos = pos

    def remove_agent(self, agent):
        x, y = agent.pos
        self._grid[x][y].remove(agent)
        agent.pos = None

    def move_agent(self, agent, pos):
        self.remove_agent(agent)
        self.place_agent(agent, pos)

    def coord_iter(self):
        for x in range(self.width):
            for y in range(self.height):
                yield self._grid[x][y], (x, y)

    def get_cell_list_contents(self, cell_list):
        out = []
        for (x, y) in cell_list:
            out.extend(self._grid[x][y])
        return out
