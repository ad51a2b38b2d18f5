#!/usr/bin/env node
/* node buildwitness.js -i witness.json -o witness.bin   (same options as the reference's tools/buildwitness.js) */
"use strict";
const fs = require("fs");
const { witnessJsonToBin } = require("../formats.js");
const a = process.argv.slice(2);
const opt = (s, l, d) => { const i = Math.max(a.indexOf(s), a.indexOf(l)); return i >= 0 && i + 1 < a.length ? a[i + 1] : d; };
fs.writeFileSync(opt("-o", "--output", "witness.bin"), witnessJsonToBin(JSON.parse(fs.readFileSync(opt("-i", "--input", "witness.json"), "utf8"))));
