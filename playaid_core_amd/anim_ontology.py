"""Action label table.

The reference derives ``MOVE_TO_CLASS_ID`` from the insertion order of
``ONTOLOGY["all"]`` (``playaid/anim_ontology.py:7-392,592-600``): 63 names,
``Jab`` = 0 ... ``Grabbed`` = 62. ``ai_runner.py:166`` passes
``list(MOVE_TO_CLASS_ID.keys())`` as the model's ``actions``, and
``ai_runner.py:475`` indexes it with the argmax. Only that ordered name table
is needed on the hot path; it is reproduced here as data.
"""

ACTIONS = [
    "Jab", "DashAttack", "ForwardTilt", "DownTilt", "UpTilt", "ForwardSmash",
    "DownSmash", "UpSmash", "NeutralSpecial", "ForwardSpecial", "DownSpecial",
    "UpSpecial", "NeutralAir", "ForwardAir", "BackAir", "DownAir", "UpAir",
    "ZAir", "Grab", "GrabRelease", "Parry", "Pummel", "ForwardThrow",
    "BackThrow", "DownThrow", "UpThrow", "Jump", "ShortHop", "Fall",
    "SpecialFall", "Shield", "ShieldStun", "ShieldDrop", "Damaged", "Wait",
    "Walk", "Squat", "Dash", "Run", "Turn", "PlatformDrop", "AirDodge", "Roll",
    "SpotDodge", "DownWait", "MissedTech", "TechInPlace", "TechRoll",
    "NormalGetUp", "GetUpAttack", "Taunt", "LedgeHang", "LedgeAttack",
    "LedgeNormalGetUp", "LedgeRoll", "LedgeJump", "LedgeGrab", "ItemPickup",
    "ItemThrow", "Slip", "Landing", "Undefined", "Grabbed",
]

MOVE_TO_CLASS_ID = {name: i for i, name in enumerate(ACTIONS)}
assert len(MOVE_TO_CLASS_ID) == 63

# fighter enums used by the ai_output -> timeline overlay (timeline.py:57-62,
# anim_ontology.py:395-492): only the two fighters that path hard-codes.
FIGHTER_NAME_TO_ENUM = {"Pikachu": 8, "Joker": 82}
FIGHTER_ENUM_TO_NAME = {v: k for k, v in FIGHTER_NAME_TO_ENUM.items()}
