"""Scene tables: the actors of the reference's two environments as plain data.

Values restate the reference's scene description files (they parameterise the kernels):
  point_env  src/m3p2i_aip/config/point_env/{0_point_robot,1..4_wall,5_obs,6_dyn_obs,7_box,
             8_goal,9_yaxis,10_xaxis}.yaml + assets/urdf/pointRobot.urdf
  panda_env  src/m3p2i_aip/config/panda_env/{1_table,2_table_stand,3_shelf_stand,4_obs,
             5_cubeA,6_cubeB,panda}.yaml + franka_panda.urdf link names

ACTOR ORDER.  The reference lists the yaml files with pathlib.iterdir() (actor_utils.py:97),
i.e. in filesystem order, and its suction model writes the robot reaction force to the LAST
rigid body of the env (skill_utils.py:89-90) -- so it only works when the robot is the last
actor.  This build fixes the order: non-robot actors in numeric file order, robot last.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Actor:
    type: str                 # "box" | "robot"
    name: str
    size: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.1])
    init_pos: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    init_ori: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0, 1.0])
    fixed: bool = False
    collision: bool = True
    friction: float = 1.0
    gravity: bool = True
    links: List[str] = field(default_factory=lambda: ["box"])
    init_joint_pose: Optional[List[float]] = None
    handle: Optional[int] = None


S = 0.707107

POINT_ENV = [
    Actor("box", "wall-1", [0.1, 8, 0.2], [4.0, 0.0, 0.0], fixed=True),
    Actor("box", "wall-2", [0.1, 8, 0.2], [-4.0, 0.0, 0.0], fixed=True),
    Actor("box", "wall-3", [0.1, 8, 0.2], [0.0, 4.0, 0.0], [0.0, 0.0, S, S], fixed=True),
    Actor("box", "wall-4", [0.1, 8, 0.2], [0.0, -4.0, 0.0], [0.0, 0.0, S, S], fixed=True),
    Actor("box", "obs", [0.3, 0.4, 0.5], [2.0, 2.0, 0.0], fixed=True),
    Actor("box", "dyn-obs", [0.4, 0.4, 0.1], [-2.0, 2.0, 0.0]),
    Actor("box", "box", [0.4, 0.4, 0.1], [0.0, 2.0, 0.0], friction=0.5),
    Actor("box", "goal", [0.45, 0.45, 0.01], [-3.75, -3.75, 0.0], fixed=True, collision=False),
    Actor("box", "yaxis", [0.05, 0.5, 0.01], [0.0, 0.25, 0.01], fixed=True, collision=False),
    Actor("box", "xaxis", [0.5, 0.05, 0.01], [0.25, 0.0, 0.01], fixed=True, collision=False),
    Actor("robot", "point_robot", init_pos=[0.0, 0.0, 0.05], fixed=True, friction=0.05,
          links=["plane", "link_x", "link_y"]),
]

PANDA_LINKS = ["panda_link0", "panda_link1", "panda_link2", "panda_link3", "panda_link4",
               "panda_link5", "panda_link6", "panda_link7", "panda_hand", "panda_leftfinger",
               "panda_rightfinger"]

PANDA_ENV = [
    Actor("box", "table", [1.2, 1.2, 0.05], [0.0, 0.0, 1.0], fixed=True),
    Actor("box", "table_stand", [0.2, 0.2, 0.1], [-0.5, 0.0, 1.075], fixed=True),
    Actor("box", "shelf_stand", [0.2, 0.2, 0.3], [0.5, 0.0, 1.175], fixed=True),
    Actor("box", "dyn-obs", [0.2, 0.2, 0.02], [0.35, 0.0, 1.735], gravity=False),
    Actor("box", "cubeA", [0.05, 0.05, 0.05], [0.2, -0.2, 1.06]),
    Actor("box", "cubeB", [0.05, 0.05, 0.05], [0.2, 0.2, 1.06]),
    Actor("robot", "panda", init_pos=[-0.45, 0.0, 1.125], fixed=True, gravity=False,
          links=PANDA_LINKS,
          init_joint_pose=[0, 0, 0, 0, 0, 0, -2, 0, 0, 0, 1.8675, 0, 0, 0, 0.02, 0, 0.02, 0]),
]
CUBE_A_ON_SHELF = [0.425, 0.0, 1.35]

ENVS = {"point_env": POINT_ENV, "panda_env": PANDA_ENV}
DOFS = {"point_env": 2, "panda_env": 9}


def actor_index(env_type: str, name: str) -> int:
    return [a.name for a in ENVS[env_type]].index(name)


def body_index(env_type: str, actor: str, link: str) -> int:
    """Index of (actor, link) among the env's rigid bodies (DOMAIN_ENV numbering)."""
    b = 0
    for a in ENVS[env_type]:
        if a.name == actor:
            return b + a.links.index(link)
        b += len(a.links)
    raise ValueError(f"unknown actor {actor!r}")


def num_bodies(env_type: str) -> int:
    return sum(len(a.links) for a in ENVS[env_type])
