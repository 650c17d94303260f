"""MI355X-native per-timestep agent-update engine (drop-in for the hot path of
kurisu-n/TrafficSimulation: CityModel.step() and everything below it)."""
__version__ = "0.1.0"
