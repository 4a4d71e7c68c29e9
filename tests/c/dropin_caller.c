/* A caller written against libbtbb's public API only (what an Ubertooth-style capture loop
 * does, lib/src/btbb.h:63-198): find access codes in a window of one-symbol-per-byte data,
 * hand the packet to the piconet logic, decode it.  Compiled as C90 against include/btbb.h and
 * linked with the drop-in library; run on the GPU box by tests/test_c_dropin.py.
 * Input file: n symbols (bytes 0/1).  Output: one line per access code. */
#include <stdio.h>
#include <stdlib.h>
#include <btbb.h>

int main(int argc, char **argv)
{
	FILE *f;
	char *syms;
	long n, off = 0;
	int r, found = 0;
	btbb_packet *pkt = NULL;
	btbb_piconet *pn;
	unsigned long lap;
	btbb_pcap_handle *pcap = NULL;
	btbb_pcapng_handle *pcapng = NULL;
	char path[512];

	if (argc < 3)
		return 2;
	lap = strtoul(argv[2], NULL, 0);
	f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	fseek(f, 0, SEEK_END);
	n = ftell(f);
	fseek(f, 0, SEEK_SET);
	syms = (char *)malloc((size_t)n + 128);
	if (fread(syms, 1, (size_t)n, f) != (size_t)n)
		return 2;
	fclose(f);

	if (btbb_init(2) < 0)
		return 3;
	pn = btbb_piconet_new();
	btbb_init_piconet(pn, (uint32_t)lap);
	if (argc > 3) {              /* capture files: <prefix>.pcap and <prefix>.pcapng */
		sprintf(path, "%.500s.pcap", argv[3]);
		if (btbb_pcap_create_file(path, &pcap) != 0)
			return 4;
		sprintf(path, "%.500s.pcapng", argv[3]);
		if (btbb_pcapng_create_file(path, "dropin_caller", &pcapng) != 0)
			return 4;
	}
	while (off < n - 64) {
		r = btbb_find_ac(syms + off, (int)(n - 63 - off), LAP_ANY, 2, &pkt);
		if (r < 0)
			break;
		off += r;
		btbb_packet_set_data(pkt, syms + off, (int)(n - off), 17, (uint32_t)(off / 312));
		printf("AC offset=%ld lap=%06x err=%u hdr=%d\n", off, (unsigned)btbb_packet_get_lap(pkt),
		       (unsigned)btbb_packet_get_ac_errors(pkt), btbb_header_present(pkt));
		if (btbb_packet_get_lap(pkt) == lap)
			btbb_process_packet(pkt, pn);
		if (pcap) {
			btbb_pcap_append_packet(pcap, (uint64_t)off * 1000u, -40, -90, (uint32_t)lap, UAP_ANY, pkt);
			btbb_pcapng_append_packet(pcapng, (uint64_t)off * 1000u, -40, -90, (uint32_t)lap, UAP_ANY, pkt);
		}
		found++;
		off += 1;
	}
	printf("DONE found=%d uap_valid=%d uap=%02x\n", found, btbb_piconet_get_flag(pn, BTBB_UAP_VALID),
	       (unsigned)btbb_piconet_get_uap(pn));
	if (pcap) {
		btbb_pcapng_record_bdaddr(pcapng, btbb_piconet_get_bdaddr(pn), 0xff, 0);
		btbb_pcap_close(pcap);
		btbb_pcapng_close(pcapng);
	}
	if (pkt)
		btbb_packet_unref(pkt);
	btbb_piconet_unref(pn);
	free(syms);
	return 0;
}
