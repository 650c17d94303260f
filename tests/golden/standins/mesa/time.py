"""RandomActivation stand-in with the SURVEY.md §8(a) A4 contract:
keys = list(_agents.keys()) in insertion order -> model.random.shuffle(keys) ->
for each key still present: agent.step()."""


class BaseScheduler:
    def __init__(self, model):
        self.model = model
        self.steps = 0
        self.time = 0
        self._agents = {}

    def add(self, agent):
        if agent.unique_id in self._agents:
            raise Exception(
                f"Agent with unique id {repr(agent.unique_id)} already added to scheduler")
        self._agents[agent.unique_id] = agent

    def remove(self, agent):
        del self._agents[agent.unique_id]

    @property
    def agents(self):
        return list(self._agents.values())

    def get_agent_count(self):
        return len(self._agents)

    def get_agent_keys(self):
        return list(self._agents.keys())

    def do_each(self, method, agent_keys=None, shuffle=False):
        if agent_keys is None:
            agent_keys = self.get_agent_keys()
        if shuffle:
            self.model.random.shuffle(agent_keys)
        for agent_key in agent_keys:
            if agent_key in self._agents:
                getattr(self._agents[agent_key], method)()
        return agent_keys

    def step(self):
        self.do_each("step")
        self.steps += 1
        self.time += 1


class RandomActivation(BaseScheduler):
    def step(self):
        self.do_each("step", shuffle=True)
        self.steps += 1
        self.time += 1
