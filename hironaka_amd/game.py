"""Autonomous games between a fixed host and a fixed agent, vectorised over the batch -- the counterpart of
``hironaka/game.py`` (`Game`, `GameHironaka`; BASELINE config 1's plumbing).

The state is a ``HipPoints`` container; one ``step()`` is: the host picks a coordinate subset per game
(`Host.select_coord`, one launch for Zeillinger's), the agent picks an axis and applies shift -> Newton
polytope in ONE fused launch (`Agent.move`, agent.py:69-72), then the optional rescale (game.py:107-108).
``coord_history`` holds the hosts' multi-binary masks [B, d] and ``move_history`` the agents' axes [B]
(-1 where the reference records ``None``), one entry per step, as device tensors.
"""
import abc
import logging
from typing import Optional

from .agent import Agent
from .core import HipPoints
from .host import Host


class Game(abc.ABC):
    """game.py:11-81"""

    def __init__(self, state: Optional[HipPoints], host: Host, agent: Agent,
                 scale_observation: Optional[bool] = True, **kwargs):
        self.logger = logging.getLogger(type(self).__name__)
        self.state = state
        self.dimension = state.dimension if state is not None else None
        self.host = host
        self.agent = agent
        self.coord_history = []
        self.move_history = []
        self.scale_observation = scale_observation
        if self.state is not None:
            self.state.get_newton_polytope()  # clear up the extra points (game.py:46)
            self.stopped = self.state.ended
            if self.scale_observation:
                self.state.rescale()
        else:
            self.stopped = True

    @abc.abstractmethod
    def step(self, verbose: int = 0) -> bool:
        """one move of every game; True if the games go on, False if they had stopped or stop now"""

    def _show(self, coords, action, weights, ended):
        lines = [f"Host move: {coords}", f"Agent move: {action}"]
        if weights is not None:
            lines.append(f"Weights: {weights}")
        for line in lines + [f"Game Ended: {ended}"]:
            self.logger.info(line)

    def print_history(self):
        for title, history in (("Coordinate history (host choices):", self.coord_history),
                               ("Move history (agent choices):", self.move_history)):
            self.logger.info(title)
            self.logger.info(history)


class GameHironaka(Game):
    """game.py:84-119"""

    def step(self, verbose: int = 0) -> bool:
        if self.stopped:
            return False
        if verbose:
            self.logger.info(self.state)
        coords = self.host.select_coord(self.state)
        action = self.agent.move(self.state, coords)
        if self.scale_observation:
            self.state.rescale()
        if verbose:
            self._show(coords, action, None, self.state.ended)
        self.coord_history.append(coords)
        self.move_history.append(action)
        if self.state.ended:
            self.stopped = True
            return False
        return True
