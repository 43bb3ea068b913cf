"""Action / fighter / stage tables of the output contract.

The reference derives ``MOVE_TO_CLASS_ID`` from the insertion order of
``ONTOLOGY["all"]`` (``playaid/anim_ontology.py:7-392,592-600``): 63 names,
``Jab`` = 0 ... ``Grabbed`` = 62. ``ai_runner.py:166`` passes
``list(MOVE_TO_CLASS_ID.keys())`` as the model's ``actions`` and
``ai_runner.py:475`` indexes it with the argmax.

``Fighter.set_from_json`` (``fighter.py:540-552``) additionally maps the logged
``motion_kind`` to an action: hex string -> param string (``HEX_TO_ACTION``, an
87 167-row CSV read at import in the reference, ``anim_ontology.py:574-578``;
here loaded on demand from a path the caller supplies, ``load_hex_to_action``)
-> action name (``PARAM_STRING_TO_ANIMATION`` + the prefix search of
``dataset_utils.py:22-60``). The tables below are the data of ``ONTOLOGY["all"]``
in a one-row-per-action form: ``"<action> <advantage><option group> <param strings>"``
with advantage n(eutral)/d(isadvantage) and option group t(ech)/l(edge)/-.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, Optional

_ACTION_ROWS = [
    "Jab n- attack_1",
    "DashAttack n- attack_dash",
    "ForwardTilt n- attack_s3",
    "DownTilt n- attack_lw3",
    "UpTilt n- attack_hi3",
    "ForwardSmash n- attack_s4",
    "DownSmash n- attack_lw4",
    "UpSmash n- attack_hi4",
    "NeutralSpecial n- special_n,special_air_n",
    "ForwardSpecial n- special_s,special_air_s",
    "DownSpecial n- special_lw,special_air_lw",
    "UpSpecial n- special_hi,special_air_hi",
    "NeutralAir n- attack_air_n",
    "ForwardAir n- attack_air_f",
    "BackAir n- attack_air_b",
    "DownAir n- attack_air_lw",
    "UpAir n- attack_air_hi",
    "ZAir n- air_catch",
    "Grab n- catch",
    "GrabRelease n- grabrelease",
    "Parry n- just_shield_off",
    "Pummel n- pummel",
    "ForwardThrow n- throw_f,throw_f_f",
    "BackThrow n- throw_b,throw_f_b",
    "DownThrow n- throw_lw,throw_f_lw",
    "UpThrow n- throw_hi,throw_f_hi",
    "Jump n- jump",
    "ShortHop n- jump_f_mini,jump_b_mini",
    "Fall n- fall",
    "SpecialFall n- specialfall",
    "Shield n- guard_on,guard_damage",
    "ShieldStun n- ",
    "ShieldDrop n- guard_off",
    "Damaged d- damage,wall_damage,thrown",
    "Wait n- wait",
    "Walk n- walk",
    "Squat n- squat",
    "Dash n- dash",
    "Run n- run",
    "Turn n- turn",
    "PlatformDrop n- pass,platform_drop",
    "AirDodge d- escape_air",
    "Roll n- escape_b,escape_f",
    "SpotDodge n- escape",
    "DownWait dt down_wait",
    "MissedTech dt down_bound",
    "TechInPlace dt passive",
    "TechRoll dt tech_roll",
    "NormalGetUp dt normalgetup",
    "GetUpAttack dt slip_attack,down_attack",
    "Taunt n- appeal",
    "LedgeHang dl cliff_wait",
    "LedgeAttack dl cliff_attack",
    "LedgeNormalGetUp dl cliff_climb",
    "LedgeRoll dl cliff_escape",
    "LedgeJump dl cliff_jump",
    "LedgeGrab dl cliff_catch",
    "ItemPickup n- item_light_get",
    "ItemThrow n- item_light_throw",
    "Slip d- slip",
    "Landing n- landing",
    "Undefined n- undefined",
    "Grabbed n- caught",
]

_FIGHTER_NAMES = (
    "Mario", "Donkey Kong", "Link", "Samus", "Dark Samus", "Yoshi",
    "Kirby", "Fox", "Pikachu", "Luigi", "Ness", "Captain Falcon",
    "Jigglypuff", "Peach", "Daisy", "Bowser", "Ice Climbers", "Sheik",
    "Zelda", "Dr. Mario", "Falco", "Marth", "Lucina", "Young Link",
    "Ganondorf", "Mewtwo", "Roy", "Chrom", "Game & Watch", "Meta Knight",
    "Pit", "Dark Pit", "Zero Suit Samus", "Wario", "Snake", "Ike",
    "Pokemon Trainer - Squirtle", "Pokemon Trainer - Ivysaur", "Pokemon Trainer - Charizard", "Diddy Kong", "Lucas", "Sonic",
    "King Dedede", "Olimar", "Lucario", "R.O.B.", "Toon Link", "Wolf",
    "Villager", "Mega Man", "Wii-Fit Trainer", "Rosalina & Luma", "Little Mac", "Greninja",
    "Palutena", "Pac-Man", "Robin", "Shulk", "Bowser Jr.", "Duck Hunt",
    "Ryu", "Ken", "Cloud", "Corrin", "Bayonetta", "Inkling",
    "Ridley", "Simon", "Richter", "King K. Rool", "Isabelle", "Incineroar",
    "??", "??", "??", "??", "??", "??",
    "??", "??", "??", "Piranha Plant", "Joker", "Hero",
    "Banjo & Kazooie", "Terry", "Byleth", "Min Min", "Steve", "Sephiroth",
    "Pyra", "Mythra", "Kazuya", "Sora",
)

ACTIONS = []
PARAM_STRING_TO_ANIMATION: Dict[str, str] = {}
MOVE_TO_ADVANTAGE_STATE: Dict[str, str] = {}
OPTION_GROUP: Dict[str, str] = {}
for _row in _ACTION_ROWS:
    _name, _flags, _params = _row.split(" ")
    ACTIONS.append(_name)
    MOVE_TO_ADVANTAGE_STATE[_name] = "disadvantage" if _flags[0] == "d" else "neutral"
    OPTION_GROUP[_name] = {"t": "tech", "l": "ledge", "-": ""}[_flags[1]]
    for _p in filter(None, _params.split(",")):
        PARAM_STRING_TO_ANIMATION[_p] = _name

MOVE_TO_CLASS_ID = {name: i for i, name in enumerate(ACTIONS)}
assert len(MOVE_TO_CLASS_ID) == 63

# anim_ontology.py:395-492 (enum 0..93; 72-80 are unassigned "??" slots)
FIGHTER_ENUM_TO_NAME = dict(enumerate(_FIGHTER_NAMES))
# the reference's inverse map is a dict comprehension over the above, so the LAST enum wins for "??"
FIGHTER_NAME_TO_ENUM = {v: k for k, v in FIGHTER_ENUM_TO_NAME.items()}

# STAGE_ENUM_TO_DATA (anim_ontology.py:497-570): fov 50 everywhere except TOWN_AND_CITY; a stage id
# outside the table falls back to stage 0 (fighter.py:479-482)
STAGE_ENUM_TO_DATA = {
    sid: {"name": name, "fov": 30 if sid == 95 else 50}
    for sid, name in (
        (0, "BATTLEFIELD"), (3, "FINAL_DESTINATION"), (44, "YOSHI_ISLAND"), (51, "FOUNTAIN_OF_DREAMS"),
        (86, "YOSHI_ISLAND_OMEGA"), (89, "HOLLOW_BASTION"), (95, "TOWN_AND_CITY"), (107, "POKEMON_STADIUM_2"),
        (118, "NEW_PORK_CITY"), (242, "KALOS"), (257, "SMASHVILLE"), (268, "PILOT_WINGS"),
        (293, "UMBRA_CLOCK_TOWER"), (295, "UMBRA_CLOCK_TOWER"), (330, "MEMENTOS"), (347, "SMALL_BATTLEFIELD"),
        (351, "NORTHERN_CAVE"), (361, "HOLLOW_BASTION"),
    )
}

# STATUS_ENUM_TO_STRING[30] == "FIGHTER_STATUS_KIND_GUARD_DAMAGE" (anim_ontology.py:636): the one status
# get_anim_for_string_and_status_kind looks at
STATUS_KIND_GUARD_DAMAGE = 30

MOTION_HEX_DIGITS = 12  # f"{motion_kind:#012x}" so that it matches the CSV's first column (fighter.py:541-542)

HEX_TO_ACTION: Dict[str, str] = {}
_hex_table_path: Optional[str] = None


def motion_hex(motion_kind: int) -> str:
    return f"{motion_kind:#0{MOTION_HEX_DIGITS}x}"


def load_hex_to_action(csv_path: Optional[str] = None) -> Dict[str, str]:
    """Fill ``HEX_TO_ACTION`` from a ``params_labels.csv`` (two columns: ``0x%010x,label``). The path
    comes from the caller or ``$PLAYAID_PARAMS_LABELS``; the reference's copy is
    ``playaid/game_data/params_labels.csv`` (``constants.py``: ``PARAMS_LABELS``). Later rows
    overwrite earlier ones with the same key, as the reference's loop does."""
    global _hex_table_path
    csv_path = csv_path or os.environ.get("PLAYAID_PARAMS_LABELS")
    if not csv_path:
        raise FileNotFoundError("no params_labels.csv: pass a path or set PLAYAID_PARAMS_LABELS")
    if csv_path != _hex_table_path:
        table = {}
        with open(csv_path, newline="") as f:
            for row in csv.reader(f, delimiter=","):
                table[row[0]] = row[1]
        HEX_TO_ACTION.clear()
        HEX_TO_ACTION.update(table)
        _hex_table_path = csv_path
    return HEX_TO_ACTION


def animation_for_param_string(param_string: str) -> str:
    """``get_animation_type_for_param_string`` (``dataset_utils.py:22-44``): exact hit, else the
    reference strips trailing characters one at a time and keeps the LAST prefix that is a key --
    i.e. the SHORTEST matching proper prefix wins ("escape_air_x" -> "escape" -> SpotDodge, not
    AirDodge) -- else "Undefined"."""
    hit = PARAM_STRING_TO_ANIMATION.get(param_string)
    if hit is not None:
        return hit
    for cut in range(1, len(param_string)):  # proper prefixes, shortest first
        hit = PARAM_STRING_TO_ANIMATION.get(param_string[:cut])
        if hit is not None:
            return hit
    return "Undefined"


def anim_for_string_and_status_kind(action_string: str, status_kind: int) -> str:
    """``get_anim_for_string_and_status_kind`` (``dataset_utils.py:47-60``)."""
    if status_kind == STATUS_KIND_GUARD_DAMAGE:
        return "ShieldStun"
    return animation_for_param_string(action_string)
