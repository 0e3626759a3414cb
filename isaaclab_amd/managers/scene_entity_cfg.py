"""``SceneEntityCfg`` (interface of ``isaaclab/managers/scene_entity_cfg.py``): a scene entity plus the joints / bodies
of it a term works on, resolved from names to indices once."""

from __future__ import annotations

from dataclasses import dataclass

from ..utils import string as string_utils


@dataclass
class SceneEntityCfg:
    name: str = "robot"
    joint_names: str | list[str] | None = None
    joint_ids: list[int] | slice = slice(None)
    body_names: str | list[str] | None = None
    body_ids: list[int] | slice = slice(None)
    preserve_order: bool = False

    def resolve(self, scene):
        """scene_entity_cfg.py:112-252: names -> ids (or ids -> names); selecting everything becomes ``slice(None)``."""
        if self.name not in scene.keys():
            raise ValueError(f"The scene entity '{self.name}' does not exist. Available entities: {scene.keys()}.")
        entity = scene[self.name]
        for names_attr, ids_attr, finder, all_names_attr, count_attr in (
                ("joint_names", "joint_ids", "find_joints", "joint_names", "num_joints"),
                ("body_names", "body_ids", "find_bodies", "body_names", "num_bodies")):
            names, ids = getattr(self, names_attr), getattr(self, ids_attr)
            if names is not None or ids != slice(None):
                if names is not None and ids != slice(None):
                    if isinstance(names, str):
                        names = [names]
                    if isinstance(ids, int):
                        ids = [ids]
                    found, _ = getattr(entity, finder)(names, preserve_order=self.preserve_order)
                    listed = [getattr(entity, all_names_attr)[i] for i in ids]
                    if found != ids or listed != names:
                        raise ValueError(f"Both '{names_attr}' and '{ids_attr}' are specified, and are not consistent."
                                         f"\n\tfrom names: {names} [{found}]\n\tfrom ids: {listed} [{ids}]")
                    setattr(self, names_attr, names), setattr(self, ids_attr, ids)
                elif names is not None:
                    if isinstance(names, str):
                        names = [names]
                    found, _ = getattr(entity, finder)(names, preserve_order=self.preserve_order)
                    setattr(self, names_attr, names)
                    # everything selected in the entity's own order -> slice(None) (scene_entity_cfg.py:171-172, 246-247)
                    everything = len(found) == getattr(entity, count_attr) and names == getattr(entity, all_names_attr)
                    setattr(self, ids_attr, slice(None) if everything else found)
                else:
                    if isinstance(ids, int):
                        ids = [ids]
                    setattr(self, ids_attr, ids)
                    setattr(self, names_attr, [getattr(entity, all_names_attr)[i] for i in ids])
