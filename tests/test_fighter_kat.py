"""The one known answer the reference's own tests hold on the output-contract side of the path
(``playaid/fighter_test.py``): ``motion_kind 19292652517 -> "wait" -> "Wait"`` plus the scalar
copy-through of ``Fighter.set_from_json`` (``playaid/fighter.py:458-555``)."""
import json
import os

import numpy as np
import pytest

from oracle import projection as oracle_projection
from playaid_core_amd import anim_ontology
from playaid_core_amd.fighter import Fighter, YoloCrop

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def kat(tmp_path):
    k = json.load(open(os.path.join(GOLD, "fighter_kat.json")))
    csv_path = tmp_path / "params_labels.csv"
    csv_path.write_text("0x0000000000,\n" + "\n".join(k["params_labels_rows"]) + "\n")
    anim_ontology.load_hex_to_action(str(csv_path))
    yield k
    anim_ontology.HEX_TO_ACTION.clear()
    anim_ontology._hex_table_path = None


def test_reference_fighter_known_answer(kat):
    data = dict(kat["data"], **kat["added_keys"])
    fighter = Fighter(frame_num=0, data=data)
    for name, want in kat["expected"].items():
        assert getattr(fighter, name) == want, name
    assert fighter.fighter_name == "Byleth" and fighter.stage == "BATTLEFIELD"
    # the box comes from the log camera (no "crop" key): same numbers as the literal numpy oracle
    want = oracle_projection.project_box(
        data["pos_x"], data["pos_y"], list(data["camera_position"].values()), list(data["camera_target_position"].values()), 50
    )
    assert fighter.crop.yolo_crop() == want
    assert 0 < fighter.crop.crop_width < 1 and 0 < fighter.crop.crop_height < 1


def test_missing_key_is_a_keyerror_like_the_reference(kat):
    with pytest.raises(KeyError):  # the stale reference test dies exactly here (fighter.py:475)
        Fighter(frame_num=0, data=dict(kat["data"]))


def test_ai_keys_override_and_update_deltas(kat):
    data = dict(kat["data"], **kat["added_keys"])
    f = Fighter(frame_num=0, data=dict(data, crop="2 0.5 0.25 0.1 0.2 0.9", action="Jab"))
    assert f.action == "Jab" and f.action_string == "wait" and str(f.crop) == "2 0.5 0.25 0.1 0.2 0.9"
    f.update(1, dict(data, damage=12.5, status_kind=30))  # GUARD_DAMAGE wins over the param string
    assert f.action == "ShieldStun" and f.previous_action == "Jab" and f.new_action and f.move_counter == 1
    assert f.damage_delta == 12.5 and f.frames_since_damaged == 0 and f.animation_frame_num == 1
    f.update(2, dict(data, damage=0.0, status_kind=30))  # respawn: damage falls, delta clamps to 0
    assert f.damage_delta == 0 and not f.new_action and f.animation_frame_num == 2 and f.frames_since_damaged == 1
    f.update(3, dict(data, motion_kind=1))  # hex the table lacks -> "" -> Undefined
    assert f.action_string == "" and f.action == "Undefined" and f.previous_non_damaged_action == "ShieldStun"


def _literal_prefix_search(key, table):
    """dataset_utils.py:22-36 as written (loop over negative slice ends, last hit wins)."""
    if key in table:
        return table[key]
    match = "Undefined"
    for i in range(0, -1 * len(key), -1):
        if key[0:i] in table:
            match = table[key[0:i]]
    return match


@pytest.mark.parametrize(
    "s", ["wait", "wait_2", "escape_air_slide", "escape_f", "attack_s4_hold", "attack_air_lw", "special_air_hi_end", "",
          "x", "zz", "cliff_jump_quick_2", "throw_f_lw", "guard_damage", "jump_b_mini", "jump_aerial_f", "passive_stand_b",
          "damage_fly_roll", "item_light_throw_air_f", "landing_air_n", "catch_wait", "caught_pulled"]
)
def test_param_string_prefix_search_matches_the_literal_loop(s):
    assert anim_ontology.animation_for_param_string(s) == _literal_prefix_search(s, anim_ontology.PARAM_STRING_TO_ANIMATION)


def test_ontology_tables():
    assert anim_ontology.FIGHTER_ENUM_TO_NAME[8] == "Pikachu" and anim_ontology.FIGHTER_ENUM_TO_NAME[82] == "Joker"
    assert anim_ontology.FIGHTER_NAME_TO_ENUM["??"] == 80  # dict comprehension: the last duplicate wins
    assert anim_ontology.STAGE_ENUM_TO_DATA[95]["fov"] == 30 and anim_ontology.STAGE_ENUM_TO_DATA[86]["fov"] == 50
    assert anim_ontology.OPTION_GROUP["TechRoll"] == "tech" and anim_ontology.OPTION_GROUP["LedgeHang"] == "ledge"
    assert anim_ontology.MOVE_TO_ADVANTAGE_STATE["Damaged"] == "disadvantage"
    assert YoloCrop.from_pixel_coordinates(100, 50, 10, 10, 30, 10, 10, 40, 30, 40).yolo_crop() == (0.2, 0.5, 0.2, 0.6)
